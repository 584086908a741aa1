/* storage_application.c — TEST INFRASTRUCTURE: CPU restatement, sequential like the reference, of
 *   InMemoryStorageTree<256, 32, 8, Blake2s256, ZkSyncStorageLeaf>     src/witness/tree/mod.rs:113-384
 *   decompose_into_storage_application_witnesses                       src/witness/individual_circuits/storage_application.rs:31-361
 *   StateDiffRecord::encode                                            circuit_encodings/src/state_diff_record.rs:21-53
 * LogQuery::derive_final_address (absent zk_evm crate, v1.4.1): Blake2s-256 of the 32-byte left-padded address
 * followed by the 32-byte big-endian key.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_blake2s256(const uint8_t *msg, size_t len, uint8_t out[32]);
void orc_keccak_f1600(uint64_t a[25]);

#define DEPTH 256

/* open-addressing map (level, masked index) -> node hash; level DEPTH holds the leaves: key -> (index, value) */
typedef struct {
    uint8_t used;
    uint16_t level;
    uint8_t key[32];
    uint8_t val[32];
    uint64_t leaf_index;
} slot_t;

struct orc_tree {
    slot_t *slots;
    size_t cap, count;
    uint64_t next_enumeration_index;
    uint8_t empty_hashes[DEPTH][32];
    uint8_t root[32];
};

static size_t slot_hash(int level, const uint8_t key[32]) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)level;
    for (int i = 0; i < 32; i++) { h ^= key[i]; h *= 1099511628211ull; }
    return (size_t)h;
}

static slot_t *tree_find(orc_tree *t, int level, const uint8_t key[32], int create) {
    if (create && (t->count + 1) * 2 > t->cap) {
        size_t ncap = t->cap * 2;
        slot_t *old = t->slots;
        size_t ocap = t->cap;
        t->slots = (slot_t *)calloc(ncap, sizeof(slot_t));
        t->cap = ncap;
        for (size_t i = 0; i < ocap; i++)
            if (old[i].used) {
                size_t p = slot_hash(old[i].level, old[i].key) & (ncap - 1);
                while (t->slots[p].used) p = (p + 1) & (ncap - 1);
                t->slots[p] = old[i];
            }
        free(old);
    }
    size_t p = slot_hash(level, key) & (t->cap - 1);
    while (t->slots[p].used) {
        if (t->slots[p].level == level && memcmp(t->slots[p].key, key, 32) == 0) return &t->slots[p];
        p = (p + 1) & (t->cap - 1);
    }
    if (!create) return NULL;
    t->slots[p].used = 1;
    t->slots[p].level = (uint16_t)level;
    memcpy(t->slots[p].key, key, 32);
    t->count++;
    return &t->slots[p];
}

static void node_hash(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]) { /* tree/mod.rs:394-402 */
    uint8_t buf[64];
    memcpy(buf, l, 32);
    memcpy(buf + 32, r, 32);
    orc_blake2s256(buf, 64, out);
}

static void leaf_hash(uint64_t index, const uint8_t value[32], uint8_t out[32]) { /* :322-329: index BE (8) || value */
    uint8_t buf[40];
    for (int b = 0; b < 8; b++) buf[b] = (uint8_t)(index >> (8 * (7 - b)));
    memcpy(buf + 8, value, 32);
    orc_blake2s256(buf, 40, out);
}

static void mask_low_bits(uint8_t idx[32], int level) { /* :187-199 */
    for (int bit = 0; bit < level; bit++) idx[bit / 8] &= (uint8_t)~(1u << (bit % 8));
}

static int is_right_side(const uint8_t idx[32], int level) { return (idx[level / 8] >> (level % 8)) & 1; }

orc_tree *orc_tree_new(void) { /* :159-198 */
    orc_tree *t = (orc_tree *)calloc(1, sizeof *t);
    t->cap = 1 << 16;
    t->slots = (slot_t *)calloc(t->cap, sizeof(slot_t));
    t->next_enumeration_index = 1;
    uint8_t zero[32] = {0}, cur[32];
    leaf_hash(0, zero, cur);
    memcpy(t->empty_hashes[0], cur, 32);
    for (int level = 1; level <= DEPTH; level++) {
        uint8_t nx[32];
        node_hash(cur, cur, nx);
        if (level < DEPTH) { memcpy(t->empty_hashes[level], nx, 32); memcpy(cur, nx, 32); }
        else memcpy(t->root, nx, 32);
    }
    return t;
}

void orc_tree_free(orc_tree *t) { if (t) { free(t->slots); free(t); } }
void orc_tree_root(const orc_tree *t, uint8_t out[32]) { memcpy(out, t->root, 32); }
uint64_t orc_tree_next_enumeration_index(const orc_tree *t) { return t->next_enumeration_index; }

static const uint8_t *path_element(orc_tree *t, int level, const uint8_t idx_in[32]) { /* :202-217 */
    uint8_t idx[32];
    memcpy(idx, idx_in, 32);
    mask_low_bits(idx, level);
    slot_t *s = tree_find(t, level, idx, 0);
    return s ? s->val : t->empty_hashes[level];
}

/* get_leaf :219-240 */
void orc_tree_get_leaf(orc_tree *t, const uint8_t key[32], uint64_t *leaf_index, uint8_t value[32], uint8_t *path /* 256*32 */) {
    slot_t *leaf = tree_find(t, DEPTH, key, 0);
    *leaf_index = leaf ? leaf->leaf_index : 0;
    if (leaf) memcpy(value, leaf->val, 32); else memset(value, 0, 32);
    for (int level = 0; level < DEPTH; level++) {
        uint8_t pair[32];
        memcpy(pair, key, 32);
        pair[level / 8] ^= (uint8_t)(1u << (level % 8));
        memcpy(path + 32 * level, path_element(t, level, pair), 32);
    }
}

/* insert_leaf :296-352; returns the leaf's enumeration index */
uint64_t orc_tree_insert_leaf(orc_tree *t, const uint8_t key[32], const uint8_t value[32], uint8_t *path /* may be NULL */) {
    slot_t *leaf = tree_find(t, DEPTH, key, 0);
    if (leaf) memcpy(leaf->val, value, 32);
    else {
        leaf = tree_find(t, DEPTH, key, 1);
        leaf->leaf_index = t->next_enumeration_index++;
        memcpy(leaf->val, value, 32);
    }
    const uint64_t index = leaf->leaf_index;
    uint8_t cur[32];
    leaf_hash(index, value, cur);
    for (int level = 0; level < DEPTH; level++) {
        uint8_t idx[32], pair[32], nx[32];
        memcpy(idx, key, 32);
        mask_low_bits(idx, level);
        memcpy(tree_find(t, level, idx, 1)->val, cur, 32);
        memcpy(pair, key, 32);
        pair[level / 8] ^= (uint8_t)(1u << (level % 8));
        const uint8_t *sib = path_element(t, level, pair);
        if (path) memcpy(path + 32 * level, sib, 32);
        if (is_right_side(key, level)) node_hash(sib, cur, nx); else node_hash(cur, sib, nx);
        memcpy(cur, nx, 32);
    }
    memcpy(t->root, cur, 32);
    return index;
}

/* verify_inclusion :242-266 */
int orc_tree_verify_inclusion(const uint8_t root[32], const uint8_t key[32], uint64_t leaf_index, const uint8_t value[32],
                              const uint8_t *path) {
    uint8_t cur[32], nx[32];
    leaf_hash(leaf_index, value, cur);
    for (int level = 0; level < DEPTH; level++) {
        if (is_right_side(key, level)) node_hash(path + 32 * level, cur, nx); else node_hash(cur, path + 32 * level, nx);
        memcpy(cur, nx, 32);
    }
    return memcmp(root, cur, 32) == 0;
}

static void u256_be(const uint32_t limbs[8], uint8_t out[32]) {
    for (int k = 0; k < 8; k++) {
        uint32_t l = limbs[7 - k];
        out[4 * k] = (uint8_t)(l >> 24); out[4 * k + 1] = (uint8_t)(l >> 16); out[4 * k + 2] = (uint8_t)(l >> 8); out[4 * k + 3] = (uint8_t)l;
    }
}

void orc_derive_final_address(const zkw_log_query *q, uint8_t out[32]) {
    uint8_t buf[64] = {0};
    for (int k = 0; k < 5; k++) {
        uint32_t l = q->address[4 - k];
        buf[12 + 4 * k] = (uint8_t)(l >> 24); buf[13 + 4 * k] = (uint8_t)(l >> 16); buf[14 + 4 * k] = (uint8_t)(l >> 8); buf[15 + 4 * k] = (uint8_t)l;
    }
    u256_be(q->key, buf + 32);
    orc_blake2s256(buf, 64, out);
}

/* StateDiffRecord::encode, state_diff_record.rs:21-53 */
void orc_state_diff_encode(const zkw_log_query *q, const uint8_t derived_key[32], uint64_t enumeration_index, uint8_t out[156]) {
    for (int k = 0; k < 5; k++) {
        uint32_t l = q->address[4 - k];
        out[4 * k] = (uint8_t)(l >> 24); out[4 * k + 1] = (uint8_t)(l >> 16); out[4 * k + 2] = (uint8_t)(l >> 8); out[4 * k + 3] = (uint8_t)l;
    }
    u256_be(q->key, out + 20);
    memcpy(out + 52, derived_key, 32);
    for (int b = 0; b < 8; b++) out[84 + b] = (uint8_t)(enumeration_index >> (8 * (7 - b)));
    u256_be(q->read_value, out + 92);
    u256_be(q->written_value, out + 124);
}

static void keccak_absorb(uint64_t st[25], const uint8_t block[136]) {
    for (int k = 0; k < 17; k++) {
        uint64_t lane = 0;
        for (int b = 0; b < 8; b++) lane |= (uint64_t)block[8 * k + b] << (8 * b);
        st[k] ^= lane;
    }
    orc_keccak_f1600(st);
}

static void encode_keccak_state(const uint64_t st[25], uint8_t out[200]) {
    for (int idx = 0; idx < 25; idx++)
        for (int b = 0; b < 8; b++) out[((idx % 5) * 5 + idx / 5) * 8 + b] = (uint8_t)(st[idx] >> (8 * b));
}

static void keccak_finalize(const uint64_t st_in[25], uint8_t out[32]) { /* hasher.clone().finalize() with an empty buffer */
    uint64_t st[25];
    memcpy(st, st_in, sizeof st);
    uint8_t block[136] = {0};
    block[0] = 0x01; block[135] = 0x80;
    keccak_absorb(st, block);
    for (int k = 0; k < 4; k++)
        for (int b = 0; b < 8; b++) out[8 * k + b] = (uint8_t)(st[k] >> (8 * b));
}

static void log_state(zkw_queue_state4 *s, const uint64_t *tails, size_t n, size_t popped) {
    memset(s, 0, sizeof *s);
    if (popped) memcpy(s->head, tails + 4 * (popped - 1), 32);
    if (n) memcpy(s->tail, tails + 4 * (n - 1), 32);
    s->length = (uint32_t)(n - popped);
}

/* Applies the deduplicated rollup storage queries to `tree` (mutated). Outputs per query: derived_keys [n][32],
   merkle_paths [n][256][32], leaf_indexes [n] (enumeration index read BEFORE the write), roots [n][32] (root after
   the query); instances sized for n + 1. Returns the number of instances or <0 when an assert of the reference
   fails (-1: read value diverges from the leaf, -2: inclusion proof does not verify). */
int64_t orc_storage_application_build(orc_tree *tree, const zkw_log_query *queries, const uint64_t *query_tails, size_t n,
                                      uint32_t capacity, uint8_t *derived_keys, uint8_t *merkle_paths, uint64_t *leaf_indexes,
                                      uint8_t *roots, zkw_storage_application_instance *instances) {
    if (capacity < 2) return -3;
    uint64_t kst[25] = {0};
    zkw_storage_application_fsm fsm_in;
    memset(&fsm_in, 0, sizeof fsm_in);
    if (n == 0) { /* :69-132 */
        zkw_storage_application_instance *o = instances;
        memset(o, 0, sizeof *o);
        o->start_flag = o->completion_flag = 1;
        const uint64_t ne = tree->next_enumeration_index;
        o->initial_next_enumeration_counter[0] = (uint32_t)ne; o->initial_next_enumeration_counter[1] = (uint32_t)(ne >> 32);
        memcpy(o->initial_root_hash, tree->root, 32);
        o->hidden_fsm_output.next_enumeration_counter[0] = (uint32_t)ne; o->hidden_fsm_output.next_enumeration_counter[1] = (uint32_t)(ne >> 32);
        memcpy(o->hidden_fsm_output.current_root_hash, tree->root, 32);
        encode_keccak_state(kst, o->hidden_fsm_output.current_diffs_keccak_accumulator_state);
        o->new_next_enumeration_counter[0] = (uint32_t)ne; o->new_next_enumeration_counter[1] = (uint32_t)(ne >> 32);
        memcpy(o->new_root_hash, tree->root, 32);
        keccak_finalize(kst, o->state_diffs_keccak256_hash);
        return 1;
    }
    /* chunking :136-165 */
    size_t *chunk_end = (size_t *)malloc((n + 1) * sizeof(size_t));
    size_t n_chunks = 0, total = 0;
    for (size_t i = 0; i < n; i++) {
        total += queries[i].rw_flag ? 2 : 1;
        if (total >= (size_t)capacity - 1) { chunk_end[n_chunks++] = i + 1; total = 0; }
    }
    if (total != 0) chunk_end[n_chunks++] = n;
    int64_t rc = (int64_t)n_chunks;
    size_t pos = 0;
    for (size_t c = 0; c < n_chunks && rc >= 0; c++) {
        zkw_storage_application_instance *o = instances + c;
        memset(o, 0, sizeof *o);
        o->start_flag = c == 0;
        o->completion_flag = c + 1 == n_chunks;
        if (c == 0) {
            const uint64_t ne = tree->next_enumeration_index;
            o->initial_next_enumeration_counter[0] = (uint32_t)ne; o->initial_next_enumeration_counter[1] = (uint32_t)(ne >> 32);
            memcpy(o->initial_root_hash, tree->root, 32);
            o->shard = 0;
            log_state(&o->storage_application_log_state, query_tails, n, 0);
        }
        o->first_item = pos;
        o->num_items = chunk_end[c] - pos;
        for (; pos < chunk_end[c]; pos++) {
            const zkw_log_query *el = queries + pos;
            uint8_t *key = derived_keys + 32 * pos, *path = merkle_paths + (size_t)32 * DEPTH * pos;
            orc_derive_final_address(el, key);
            uint64_t idx;
            uint8_t value[32], expect[32];
            orc_tree_get_leaf(tree, key, &idx, value, path);
            u256_be(el->read_value, expect);
            if (memcmp(expect, value, 32) != 0) { rc = -1; break; }
            if (!el->rw_flag && !orc_tree_verify_inclusion(tree->root, key, idx, value, path)) { rc = -2; break; }
            leaf_indexes[pos] = idx;
            if (el->rw_flag) {
                uint8_t nv[32], wpath[32 * DEPTH];
                u256_be(el->written_value, nv);
                const uint64_t nidx = orc_tree_insert_leaf(tree, key, nv, wpath);
                if (memcmp(wpath, path, sizeof wpath) != 0) { rc = -4; break; } /* :232 */
                if (!orc_tree_verify_inclusion(tree->root, key, nidx, nv, wpath)) { rc = -2; break; }
                uint8_t ext[272] = {0};
                orc_state_diff_encode(el, key, idx, ext); /* the index BEFORE writing, :239-246 */
                keccak_absorb(kst, ext);
                keccak_absorb(kst, ext + 136);
            }
            memcpy(roots + 32 * pos, tree->root, 32);
        }
        if (rc < 0) break;
        zkw_storage_application_fsm out;
        memset(&out, 0, sizeof out);
        const uint64_t ne = tree->next_enumeration_index;
        out.next_enumeration_counter[0] = (uint32_t)ne; out.next_enumeration_counter[1] = (uint32_t)(ne >> 32);
        memcpy(out.current_root_hash, tree->root, 32);
        log_state(&out.current_storage_application_log_state, query_tails, n, pos);
        encode_keccak_state(kst, out.current_diffs_keccak_accumulator_state);
        if (o->completion_flag) {
            o->new_next_enumeration_counter[0] = (uint32_t)ne; o->new_next_enumeration_counter[1] = (uint32_t)(ne >> 32);
            memcpy(o->new_root_hash, tree->root, 32);
            keccak_finalize(kst, o->state_diffs_keccak256_hash);
        }
        o->hidden_fsm_input = fsm_in;
        o->hidden_fsm_output = out;
        fsm_in = out;
    }
    free(chunk_end);
    return rc;
}
