"""One production-capacity block through zkw_block_run (builders) + zkw_block_synthesize, against the oracle in the
reference's order: spans, wall times, bit-exact public inputs."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic
from oracle import block as ob, pyoracle as o

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = synthetic.block_production(seed)
# pre-block storage tree from the oracle's own dedup (outside any timed region)
T = {}
t0 = time.perf_counter(); a = ob.create_artifacts_after_vm(b, timings=T); cpu_build = time.perf_counter() - t0
dedup = a["witnesses"]["storage_sorter"]["result_q"]
tree = o.Tree()
for q in dedup:
    if q["read_value"].any():
        tree.insert_leaf(o.derive_final_address(q), b"".join(int(x).to_bytes(4, "big") for x in q["read_value"][::-1]))
root0, next0 = tree.root, tree.next_enumeration_index
def answers(q):
    idx = np.zeros(q.size, np.uint64); paths = np.zeros((q.size, 256, 32), np.uint8)
    for i in range(q.size):
        idx[i], _, paths[i] = tree.get_leaf(o.derive_final_address(q[i]))
    return idx, paths
for rep in range(3):
    t0 = time.perf_counter()
    B = nv.Block(0, b, None, storage_tree=answers, storage_initial_root=root0, storage_next_enumeration_index=next0)
    t1 = time.perf_counter()
    n = B.synthesize(1 << 20, ring_slots=2)
    t2 = time.perf_counter()
    n = B.synthesize(1 << 20, ring_slots=2)
    t3 = time.perf_counter()
    print(f"rep {rep}: builders {1e3*(t1-t0):.1f} ms, synthesis of {n} instances {1e3*(t2-t1):.1f} ms (first, allocates the ring) / {1e3*(t3-t2):.1f} ms")
    if rep < 2: B.free()
for name, s, e in sorted(B.timings(), key=lambda x: x[1]):
    print(f"  {name:28s} {s:9.1f} -> {e:9.1f} ms  ({e-s:8.1f})")
print("instances:", {t: B.num_instances(t) for t in range(2, 14)})
for t, pi in a["public_inputs"].items():
    assert np.array_equal(B.public_inputs(t), pi), t
    assert np.array_equal(B.recursion_queue(t)[1], a["recursion_queues"][t][1]), t
assert B.memory_queue_state().tobytes() == a["memory_queue_state"].tobytes()
print("public inputs, recursion queues, memory queue state: bit-exact vs oracle")
bad_total = 0
def cb(ctype, inst, trace, slot, pi):
    global bad_total
    bad, first = B.check_satisfied(ctype, trace, slot)
    bad_total += bad
n = B.synthesize(1 << 20, ring_slots=2, callback=cb)
print("check_if_satisfied over", n, "instances:", bad_total, "violations")
print("cpu oracle builders (reference order, 1 thread): %.2f s" % cpu_build, {k: round(v, 2) for k, v in T.items()})
