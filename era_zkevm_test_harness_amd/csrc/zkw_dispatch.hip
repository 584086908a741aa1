// zkw_dispatch.hip — the type-dispatching synthesis entry points: the counterpart of the reference's
// `ZkSyncBaseLayerCircuit::synthesis` / `synthesis_inner` (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:286-323),
// which `match`es over the enum's thirteen variants, and of `base_test_circuit`'s `check_if_satisfied` (src/tests/mod.rs:130-259).
// Host code written against include/zkw.h only: one `switch` over BaseLayerCircuitType in front of the per-type functions, so that
// a host holds ONE function pointer per operation, like the reference's enum method. No kernels.
#include <vector>

#include "../../include/zkw.h"
#include "zkw_internal.h"

extern "C" int zkw_synthesize(zkw_ctx* ctx, uint8_t circuit_type, const void* witness, size_t first_instance, size_t n_instances, zkw_trace* t,
                              size_t first_slot) {
    if (!ctx || !witness || !t) return zkw_fail(ZKW_ERR_INVALID, "zkw_synthesize: null argument");
    void* w = const_cast<void*>(witness);
    switch (circuit_type) {
        case ZKW_CIRCUIT_CODE_DECOMMITTMENTS_SORTER: return zkw_decommit_sorter_synthesize(ctx, static_cast<const zkw_decommit_witness*>(witness), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_CODE_DECOMMITTER: return zkw_code_decommitter_synthesize(ctx, static_cast<zkw_decommitter_witness*>(w), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_LOG_DEMUXER: return zkw_log_demux_synthesize(ctx, static_cast<const zkw_demux_witness*>(witness), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_KECCAK256_ROUND_FUNCTION: return zkw_keccak_round_synthesize(ctx, static_cast<zkw_precompile_witness*>(w), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_SHA256_ROUND_FUNCTION: return zkw_sha256_round_synthesize(ctx, static_cast<zkw_precompile_witness*>(w), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_RAM_PERMUTATION: return zkw_ram_synthesize(ctx, static_cast<const zkw_ram_witness*>(witness), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_STORAGE_SORTER: return zkw_storage_sorter_synthesize(ctx, static_cast<const zkw_storage_witness*>(witness), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_STORAGE_APPLICATION: return zkw_storage_application_synthesize(ctx, static_cast<zkw_storage_application_witness*>(w), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_EVENTS_SORTER:
        case ZKW_CIRCUIT_L1_MESSAGES_SORTER: return zkw_events_sorter_synthesize(ctx, static_cast<const zkw_events_witness*>(witness), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_L1_MESSAGES_HASHER: {
            // one instance per queue (data_hasher_and_merklizer.rs:34-60): "instance k" of the witness is its k-th queue
            const zkw_linear_hasher_witness* lw = static_cast<const zkw_linear_hasher_witness*>(witness);
            if (first_instance + n_instances > lw->n_queues || !lw->message_offsets) return zkw_fail(ZKW_ERR_INVALID, "zkw_synthesize: L1MessagesHasher queues [%zu, %zu) of %zu", first_instance, first_instance + n_instances, lw->n_queues);
            // the batch entry point wants offsets that start at 0 and a records array: rebase the window [first_instance, + n_instances)
            std::vector<uint64_t> off(n_instances + 1);
            const uint64_t base = lw->message_offsets[first_instance];
            for (size_t k = 0; k <= n_instances; k++) off[k] = lw->message_offsets[first_instance + k] - base;
            std::vector<zkw_linear_hasher_instance> scratch_records(lw->records_out ? 0 : n_instances);
            return zkw_linear_hasher_synthesize_batch_with_tails(ctx, lw->messages ? lw->messages + base : nullptr, off.data(), n_instances, lw->queue_states + first_instance,
                                                                 lw->message_tails ? lw->message_tails + 4 * base : nullptr, lw->capacity, t, first_slot,
                                                                 lw->records_out ? lw->records_out + first_instance : scratch_records.data(),
                                                                 lw->public_inputs_out ? lw->public_inputs_out + 4 * first_instance : nullptr);
        }
        case ZKW_CIRCUIT_ECRECOVER: return zkw_ecrecover_synthesize(ctx, static_cast<zkw_precompile_witness*>(w), first_instance, n_instances, t, first_slot);
        case ZKW_CIRCUIT_MAIN_VM: return zkw_fail(ZKW_ERR_INVALID, "zkw_synthesize: MainVM (type 1) has no synthesis in this library (it needs the VM; DESIGN.md)");
        default: return zkw_fail(ZKW_ERR_INVALID, "zkw_synthesize: unknown circuit type %u", circuit_type);
    }
}

extern "C" int zkw_check_satisfied(zkw_ctx* ctx, uint8_t circuit_type, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations,
                                   uint64_t* first_bad) {
    switch (circuit_type) {
        case ZKW_CIRCUIT_CODE_DECOMMITTMENTS_SORTER: return zkw_decommit_sorter_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_CODE_DECOMMITTER: return zkw_code_decommitter_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_LOG_DEMUXER: return zkw_log_demux_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_KECCAK256_ROUND_FUNCTION: return zkw_keccak_round_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_SHA256_ROUND_FUNCTION: return zkw_sha256_round_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_ECRECOVER: return zkw_ecrecover_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_RAM_PERMUTATION: return zkw_ram_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_STORAGE_SORTER: return zkw_storage_sorter_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_STORAGE_APPLICATION: return zkw_storage_application_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_EVENTS_SORTER:
        case ZKW_CIRCUIT_L1_MESSAGES_SORTER: return zkw_events_sorter_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_L1_MESSAGES_HASHER: return zkw_linear_hasher_check_satisfied(ctx, t, slot, capacity, n_violations, first_bad);
        case ZKW_CIRCUIT_MAIN_VM: return zkw_fail(ZKW_ERR_INVALID, "zkw_check_satisfied: MainVM (type 1) has no layout in this library");
        default: return zkw_fail(ZKW_ERR_INVALID, "zkw_check_satisfied: unknown circuit type %u", circuit_type);
    }
}
