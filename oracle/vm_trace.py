"""TEST INFRASTRUCTURE — CPU restatement (literal, sequential, small inputs), never imported by the product.

The pre-builder half of `create_artifacts_from_tracer`: everything between the tracer's raw records and the per-circuit
builders that is NOT a builder —
  CallstackWithAuxData::{push_entry, pop_entry, add_log_query}   src/witness/callstack_handler.rs:174-460
  the log-queue forward / rollback chain with frame markers       src/witness/oracle.rs:233-499
  the callstack replay (rollback heads / tails, sponge states)    src/witness/oracle.rs:501-843
The Rust containers are kept as their Python twins (HashMap -> dict, BTreeMap -> dict + sorted keys, Vec -> list) and the
statements follow the reference in order, asserts included (AssertionError = the reference's panic). Hashing goes through
the C oracle (pyoracle.queue_push_chain_log = LogQueueSimulator pushes, pyoracle.callstack_simulate = CallstackSimulator).

Input = what WitnessTracer feeds CallstackWithAuxData (tracer.rs:221-407), as plain arrays: `events` in time order, kind 0 =
add_log_query(cycle, log_queries[index]), 1 = push_entry(cycle, entries[2*index], entries[2*index+1]), 2 = pop_entry(cycle,
panicked). The first event is the push of `from_initial_callstack` (callstack_handler.rs:163-172)."""
import bisect

import numpy as np

from . import pyoracle

EVENT = np.dtype([("kind", "<u4"), ("cycle", "<u4"), ("panicked", "<u4"), ("index", "<u4")])
LOG, PUSH, POP = 0, 1, 2
STORAGE_AUX_BYTE, EVENT_AUX_BYTE, L1_MESSAGE_AUX_BYTE, PRECOMPILE_AUX_BYTE = 0, 1, 2, 3


def _marker(kind, unique_query_id, in_frame, index, cycle):
    return (kind, unique_query_id, in_frame, index, cycle)  # kind: 'F' Forward, 'N' ForwardNoRollback, 'R' Rollback


class CallstackWithAuxData:
    """callstack_handler.rs:113-460 (only what create_artifacts_from_tracer reads)"""

    def __init__(self):  # empty(), :132-161
        self.monotonic_frame_counter = 1
        self.rollbackable_monotonic_counter = 0
        self.non_rollbackable_monotonic_counter = 0
        self.unique_query_id_counter = 0
        self.current_entry = {"history": {"action": ("Fresh",), "affected_entry": None, "frame_index": 0, "beginning_cycle": 0, "end_cycle": None},
                              "frame_index": 0, "parent_frame_index": 0, "forward_queue": [("FH", 0)], "rollback_queue": [("RT", 0)]}
        self.depth = 0
        self.stack = []
        self.full_history = [dict(self.current_entry["history"])]
        self.log_access_history = []
        self.flat_new_frames_history = []

    def push_entry(self, cycle, previous_simple_entry, new_simple_entry):  # :174-222
        self.flat_new_frames_history.append((cycle, new_simple_entry))
        new_counter = self.monotonic_frame_counter
        self.monotonic_frame_counter += 1
        self.depth += 1
        current_frame_index = self.current_entry["frame_index"]
        full_entry = {"history": {"action": ("Fresh",), "affected_entry": new_simple_entry, "frame_index": new_counter, "beginning_cycle": cycle, "end_cycle": None},
                      "parent_frame_index": current_frame_index, "frame_index": new_counter, "forward_queue": [("FH", new_counter)], "rollback_queue": [("RT", new_counter)]}
        history_of_new = dict(full_entry["history"])
        current, self.current_entry = self.current_entry, full_entry
        current["history"]["affected_entry"] = previous_simple_entry
        current["history"]["end_cycle"] = cycle
        history_of_current = dict(current["history"])
        history_of_current["action"] = ("PushToStack",)
        self.stack.append(current)
        self.full_history.append(history_of_current)
        self.full_history.append(history_of_new)

    def pop_entry(self, cycle, panicked):  # :224-346
        previous = self.stack.pop()
        self.depth -= 1
        previous["history"]["beginning_cycle"] = cycle
        previous["history"]["end_cycle"] = None
        previous_history_record = dict(previous["history"])
        previous_history_record["action"] = ("PopFromStack", panicked)
        current, self.current_entry = self.current_entry, previous
        frame_index = current["frame_index"]
        history_of_current = dict(current["history"])
        forward_queue, rollback_queue = current["forward_queue"], current["rollback_queue"]
        if panicked:
            self.current_entry["forward_queue"].extend(forward_queue)
            self.current_entry["forward_queue"].append(("FT", frame_index))
            rollback_queue.append(("RH", frame_index))
            self.current_entry["forward_queue"].extend(reversed(rollback_queue))
        else:
            self.current_entry["forward_queue"].extend(forward_queue)
            self.current_entry["forward_queue"].append(("FT", frame_index))
            self.current_entry["rollback_queue"].extend(rollback_queue)
            self.current_entry["rollback_queue"].append(("RH", frame_index))
        history_of_current["action"] = ("Exited", panicked)
        history_of_current["end_cycle"] = cycle
        self.full_history.append(history_of_current)
        self.full_history.append(previous_history_record)

    def add_log_query(self, cycle, log_index, rw_flag):  # :348-459
        current_frame_index = self.current_entry["frame_index"]
        unique_query_id = self.unique_query_id_counter
        self.unique_query_id_counter += 1
        if rw_flag:
            query_index = self.rollbackable_monotonic_counter
            self.rollbackable_monotonic_counter += 1
            marker = _marker("F", unique_query_id, current_frame_index, query_index, cycle)
            self.current_entry["forward_queue"].append(("Q", marker, cycle, log_index, False))
            self.log_access_history.append((cycle, marker))
            unique_query_id = self.unique_query_id_counter
            self.unique_query_id_counter += 1
            marker = _marker("R", unique_query_id, current_frame_index, query_index, cycle)
            self.current_entry["rollback_queue"].append(("Q", marker, cycle, log_index, True))
            self.log_access_history.append((cycle, marker))
        else:
            query_index = self.non_rollbackable_monotonic_counter
            self.non_rollbackable_monotonic_counter += 1
            marker = _marker("N", unique_query_id, current_frame_index, query_index, cycle)
            self.current_entry["forward_queue"].append(("Q", marker, cycle, log_index, False))
            self.log_access_history.append((cycle, marker))


def replay_events(events, log_queries, entries):
    ev = np.ascontiguousarray(events, dtype=EVENT)
    assert ev.size and int(ev[0]["kind"]) == PUSH, "the trace starts with from_initial_callstack's push"
    cs = CallstackWithAuxData()
    for e in ev:
        k, cycle, idx = int(e["kind"]), int(e["cycle"]), int(e["index"])
        if k == LOG:
            q = log_queries[idx]
            assert not q["rollback"]
            cs.add_log_query(cycle, idx, bool(q["rw_flag"]))
        elif k == PUSH:
            cs.push_entry(cycle, 2 * idx, 2 * idx + 1)
        else:
            cs.pop_entry(cycle, bool(e["panicked"]))
    return cs


def _range(sorted_keys, lo, hi_inclusive):
    """BTreeMap::range(lo..=hi)"""
    return sorted_keys[bisect.bisect_left(sorted_keys, lo):bisect.bisect_right(sorted_keys, hi_inclusive)]


def create_artifacts_before_builders(events, log_queries, entries):
    """oracle.rs:233-843. Returns a dict of plain lists / arrays named after the reference's locals."""
    log_queries = np.ascontiguousarray(log_queries, dtype=pyoracle.LOG_QUERY)
    entries = np.ascontiguousarray(entries, dtype=pyoracle.CALLSTACK_ENTRY)
    cs = replay_events(events, log_queries, entries)
    assert cs.depth == 0, "parent frame didn't exit"  # :236-239
    forward = cs.current_entry["forward_queue"]
    rollbacks = cs.current_entry["rollback_queue"]

    # ---- :308-499 the flattened queue: forward, then the rollbacks in reverse; hash it (LogQueueSimulator)
    flat = [(x, True) for x in forward] + [(x, False) for x in reversed(rollbacks)]
    items = [(x, applied) for x, applied in flat if x[0] == "Q"]
    fq = np.zeros(len(items), pyoracle.LOG_QUERY)
    for i, (x, _) in enumerate(items):
        fq[i] = log_queries[x[3]]
        fq[i]["rollback"] = 1 if x[4] else 0
    old_tails, new_tails = pyoracle.queue_push_chain_log(pyoracle.encode_log_queries(fq)) if len(items) else (np.zeros((0, 4), np.uint64),) * 2

    log_position_mapping = {}
    chain_of_states = []  # (cycle, marker, (previous_tail, tail))
    original_len = None  # original_log_queue_simulator = the queue when the first not-applied element is met (:322-330)
    seen_keys = set()  # sponges_data keys (:364-405)
    cycle_into_flat_sequence_index = {}
    demuxed = {k: [] for k in ("rollup_storage", "porter_storage", "event", "to_l1", "precompile")}
    original_log_queue_states = []  # (cycle, pointer)
    for x, was_applied in flat:
        if not was_applied:
            if original_len is None:
                original_len = len(chain_of_states)
        else:
            assert original_len is None  # "check for no gaps"
        if x[0] != "Q":
            log_position_mapping[x] = len(chain_of_states) - 1
            continue
        _, marker, cycle, log_index, is_rollback = x
        pointer = len(chain_of_states)
        chain_of_states.append((cycle, marker, (old_tails[pointer], new_tails[pointer])))
        q = fq[pointer]
        key = int(q["timestamp"])
        if q["rollback"]:
            assert key in seen_keys, "rollbacks always happen after forward case"
            assert cycle in cycle_into_flat_sequence_index
            cycle_into_flat_sequence_index[cycle][1] = pointer
            assert marker[0] == "R" and marker[4] == cycle
        else:
            seen_keys.add(key)
            cycle_into_flat_sequence_index.setdefault(cycle, [0, None])[0] = pointer
            assert marker[0] in ("F", "N") and marker[4] == cycle
        if was_applied:
            original_log_queue_states.append((cycle, pointer))
            aux = int(q["aux_byte"])
            if aux == STORAGE_AUX_BYTE:
                assert int(q["shard_id"]) in (0, 1)
                demuxed["rollup_storage" if int(q["shard_id"]) == 0 else "porter_storage"].append(pointer)
            elif aux == L1_MESSAGE_AUX_BYTE:
                demuxed["to_l1"].append(pointer)
            elif aux == EVENT_AUX_BYTE:
                demuxed["event"].append(pointer)
            elif aux == PRECOMPILE_AUX_BYTE:
                assert not q["rollback"]
                demuxed["precompile"].append(pointer)
            else:
                raise AssertionError("unreachable aux byte")
    if original_len is None:
        original_len = len(chain_of_states)

    # ---- :276-299 beginnings of frames
    global_beginnings_of_frames = {}
    for el in cs.full_history:
        if el["action"] == ("Fresh",):
            global_beginnings_of_frames[el["frame_index"]] = el["beginning_cycle"]
        elif el["action"][0] == "Exited":
            assert el["end_cycle"] is not None, "frame must end"
    global_beginnings_of_frames[0] = 0

    # ---- :526-563 rollback tails of new frames
    zero4 = np.zeros(4, np.uint64)
    global_end_of_storage_log = chain_of_states[-1][2][1] if chain_of_states else zero4
    frame_rollback_tails = {}
    rollback_queue_initial_tails_for_new_frames = []
    for frame_index in range(cs.monotonic_frame_counter):
        if frame_index == 0:
            tail = global_end_of_storage_log
        else:
            pos = log_position_mapping[("RT", frame_index)]
            tail = global_end_of_storage_log if pos == -1 else chain_of_states[pos][2][1]
        frame_rollback_tails[frame_index] = tail
        rollback_queue_initial_tails_for_new_frames.append((global_beginnings_of_frames[frame_index], tail))

    # ---- :571-578 rollback head segments (BTreeMap order = ascending cycle)
    cycles_sorted = sorted(cycle_into_flat_sequence_index)
    rollback_queue_head_segments = [(c, chain_of_states[cycle_into_flat_sequence_index[c][1]][2][0]) for c in cycles_sorted
                                    if cycle_into_flat_sequence_index[c][1] is not None]

    # ---- :580-843 the callstack replay
    history_of_storage_log_states = {}
    cur = {"frame_idx": 0, "forward_tail": zero4, "forward_length": 0, "rollback_head": global_end_of_storage_log,
           "rollback_tail": global_end_of_storage_log, "rollback_length": 0}
    storage_logs_states_stack = []
    state_to_merge = None
    ops, pushed = [], []  # the CallstackSimulator's operations, hashed in one go below
    witness_cycles = []   # callstack_values_witnesses[k].0
    range_cycles = [0]    # callstack_sponge_encoding_ranges[k].0
    eq4 = lambda a, b: np.array_equal(a, b)

    def same(a, b):
        return all((eq4(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k]) for k in a)

    def walk_span(begin_at_cycle, end_cycle):
        for cycle in _range(cycles_sorted, begin_at_cycle + 1, end_cycle):
            fwd, rb = cycle_into_flat_sequence_index[cycle]
            new_forward_tail = chain_of_states[fwd][2][1]
            if not eq4(new_forward_tail, cur["forward_tail"]):
                cur["forward_tail"] = new_forward_tail
                cur["forward_length"] += 1
            if rb is not None:
                cur["rollback_head"] = chain_of_states[rb][2][0]
                cur["rollback_length"] += 1
            previous = history_of_storage_log_states.get(cycle)
            history_of_storage_log_states[cycle] = dict(cur)
            if previous is not None:
                assert same(previous, cur), f"duplicate divergence for cycle {cycle}"

    for el in cs.full_history:
        frame_index = el["frame_index"]
        act = el["action"]
        if act == ("PushToStack",):
            end_cycle = el["end_cycle"]
            assert end_cycle is not None, "frame must end"
            walk_span(el["beginning_cycle"], end_cycle)
            entry = entries[el["affected_entry"]].copy()
            entry["rollback_queue_head"] = cur["rollback_head"]
            entry["rollback_queue_tail"] = cur["rollback_tail"]
            entry["rollback_queue_segment_length"] = cur["rollback_length"]
            storage_logs_states_stack.append(dict(cur))
            ops.append(1)
            pushed.append(entry)
            assert not witness_cycles or witness_cycles[-1] != end_cycle
            witness_cycles.append(end_cycle)
            range_cycles.append(end_cycle)
        elif act[0] == "PopFromStack":
            panic = act[1]
            assert state_to_merge is not None
            claimed_panic, merge = state_to_merge
            state_to_merge = None
            assert panic == claimed_panic
            popped_state = storage_logs_states_stack.pop()
            ops.append(0)  # the popped entry's rollback head / tail / length equal popped_state's by construction (:690-704)
            cur = dict(popped_state)
            cur["frame_idx"] = frame_index
            cur["forward_tail"] = merge["forward_tail"]
            assert cur["forward_length"] <= merge["forward_length"], f"divergence at frame {frame_index}"
            cur["forward_length"] = merge["forward_length"]
            if panic:
                assert eq4(cur["forward_tail"], merge["rollback_head"]), f"divergence at frame {frame_index} with panic"
                cur["forward_tail"] = merge["rollback_tail"]
                cur["forward_length"] += merge["rollback_length"]
            else:
                assert eq4(cur["rollback_head"], merge["rollback_tail"]), f"divergence at frame {frame_index} without panic"
                cur["rollback_head"] = merge["rollback_head"]
                cur["rollback_length"] += merge["rollback_length"]
            beginning_cycle = el["beginning_cycle"]
            previous = history_of_storage_log_states.get(beginning_cycle)
            history_of_storage_log_states[beginning_cycle] = dict(cur)
            if previous is not None:
                assert same(previous, cur), f"duplicate divergence for cycle {beginning_cycle}"
            assert not witness_cycles or witness_cycles[-1] != beginning_cycle
            witness_cycles.append(beginning_cycle)
            range_cycles.append(beginning_cycle)
        elif act == ("Fresh",):
            rollback_tail = frame_rollback_tails[frame_index]
            cur["frame_idx"] = frame_index
            cur["rollback_length"] = 0
            cur["rollback_head"] = rollback_tail
            cur["rollback_tail"] = rollback_tail
            cycle = el["beginning_cycle"]
            previous = history_of_storage_log_states.get(cycle)
            history_of_storage_log_states[cycle] = dict(cur)
            if previous is not None:
                assert previous["frame_idx"] < cur["frame_idx"], f"frame divergence for cycle {cycle}"
                assert eq4(previous["forward_tail"], cur["forward_tail"]) and previous["forward_length"] == cur["forward_length"]
        else:  # Exited
            assert state_to_merge is None
            end_cycle = el["end_cycle"]
            assert end_cycle is not None, "frame must end"
            walk_span(el["beginning_cycle"], end_cycle)
            state_to_merge = (act[1], dict(cur))

    sim = pyoracle.callstack_simulate(np.array(ops, np.uint8), np.array(pushed, pyoracle.CALLSTACK_ENTRY) if pushed else np.zeros(0, pyoracle.CALLSTACK_ENTRY))
    pushed_arr = np.array(pushed, pyoracle.CALLSTACK_ENTRY) if pushed else np.zeros(0, pyoracle.CALLSTACK_ENTRY)
    hist_cycles = sorted(history_of_storage_log_states)
    return {
        "flat_queries": fq, "flat_cycles": np.array([c[0] for c in chain_of_states], np.uint32),
        "flat_frames": np.array([c[1][2] for c in chain_of_states], np.uint32),
        "flat_old_tails": old_tails, "flat_new_tails": new_tails, "original_log_queue_length": original_len,
        "demuxed": demuxed, "original_log_queue_states": original_log_queue_states,
        "global_end_of_storage_log": np.array(global_end_of_storage_log, np.uint64),
        "rollback_queue_initial_tails_for_new_frames": rollback_queue_initial_tails_for_new_frames,
        "rollback_queue_head_segments": rollback_queue_head_segments,
        "history_of_storage_log_states": [(c, history_of_storage_log_states[c]) for c in hist_cycles],
        "callstack_values_witnesses": {"cycles": np.array(witness_cycles, np.uint32), "is_push": np.array(ops, np.uint8),
                                       "entries": pushed_arr[sim["entry_index"]] if len(ops) else pushed_arr, **sim},
        "callstack_sponge_encoding_ranges": (np.array(range_cycles, np.uint32),
                                             np.concatenate([np.zeros((1, 12), np.uint64), sim["new_state"]]) if len(ops) else np.zeros((1, 12), np.uint64)),
        "flat_new_frames_history": [(c, e) for c, e in cs.flat_new_frames_history],
        "monotonic_frame_counter": cs.monotonic_frame_counter,
    }
