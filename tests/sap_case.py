"""A storage-application scenario for the tests: deduplicated rollup storage queries over a tree that already holds some of
the slots, and what that tree answers for the state before the block (what zkw_storage_application_build takes)."""
import numpy as np

from era_zkevm_test_harness_amd import synthetic


def be32(limbs):
    return b"".join(int(x).to_bytes(4, "big") for x in limbs[::-1])


def storage_application_case(oracle, n, seed):
    q, existing = synthetic.storage_application_trace(n, seed=seed)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(q))[1] if n else np.zeros((0, 4), np.uint64)
    tree = oracle.Tree()
    rng = np.random.default_rng(seed)
    for _ in range(30):
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    keys = [oracle.derive_final_address(q[i]) for i in range(n)]
    for i in np.nonzero(existing)[0]:
        tree.insert_leaf(keys[i], be32(q["read_value"][i]))
    idx = np.zeros(n, np.uint64)
    paths = np.zeros((n, 256, 32), np.uint8)
    for i in range(n):
        idx[i], _, paths[i] = tree.get_leaf(keys[i])
    return q, tails, tree, idx, paths
