// netlist_eval.cuh — what the lookup tables of the netlist circuits compute (the contents of boojum's create_xor8_table, create_and8_table,
// create_byte_split_table<K>, create_tri_xor_table, create_ch4_table, create_maj4_table, create_4bit_chunk_split_table<K>; the crate is
// absent, the functions are the standard ones of the table names) and which row of the stacked tables a lookup hits. Library only: the
// test oracle has its own enumeration of the tables (oracle/netlist_tables.c) and does not include this file.
#pragma once
#include "../../include/zkw_netlist.h"

#if defined(__HIPCC__)
#define NL_HD __host__ __device__ static inline
#else
#define NL_HD static inline
#endif

/* out[0..n_out) of a table for inputs a[0..n_in) (contents of boojum's create_*_table) */
NL_HD void nl_table_eval(uint32_t fn, uint32_t k, const uint32_t a[3], uint32_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    switch (fn) {
        case NL_FN_XOR8: out[0] = a[0] ^ a[1]; break;
        case NL_FN_AND8: out[0] = a[0] & a[1]; break;
        case NL_FN_BYTESPLIT: out[0] = a[0] & ((1u << k) - 1); out[1] = a[0] >> k; break;
        case NL_FN_TRIXOR4: out[0] = a[0] ^ a[1] ^ a[2]; break;
        case NL_FN_CH4: out[0] = (a[0] & a[1]) ^ (~a[0] & a[2] & 15u); break;
        case NL_FN_MAJ4: out[0] = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]); break;
        case NL_FN_SPLIT4: out[0] = a[0] & ((1u << k) - 1); out[1] = a[0] >> k; out[2] = (out[0] << (4 - k)) | out[1]; break;
        default: break;
    }
}
/* row of the stacked table (= of the multiplicity column) that a lookup with these inputs hits */
NL_HD uint32_t nl_table_key(const nl_table *t, const uint32_t a[3]) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < t->n_in; i++) k |= a[i] << (t->in_bits * i);
    return t->offset + k;
}

