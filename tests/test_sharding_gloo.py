"""Multi-rank path on CPU: world_size 2 over gloo, each rank drives an engine (the oracle engine here;
the HIP engine on a GPU node) over its round-robin share of the batches.  The merged statistics and the
stitched result records must equal a single-rank run bit for bit — including the k-mer dictionary's
insertion order, which only works because the time keys carry the global read index."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from afterqc_amd import capi, synth


def _cfg():
    cfg = capi.Config()
    cfg.paired = 1
    cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
    cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
    cfg.barcode_length = 12
    cfg.set_verify("CAGTA")
    cfg.qc_kmer = 8
    return cfg


def _batches(n=3000, per=500):
    d = synth.make_pairs(n, 150, seed=2718, dirty=True)
    out = []
    for bi, a in enumerate(range(0, n, per)):
        b = capi.Batch.from_matrices(d["seq1"][a:a + per], d["qual1"][a:a + per], d["len1"][a:a + per],
                                     d["seq2"][a:a + per], d["qual2"][a:a + per], d["len2"][a:a + per], first_index=a)
        out.append((bi, b))
    return out


def owner(batch_index, world):
    """rank that processes batch `batch_index`: fixed-size batches dealt round robin, each carrying its first_index"""
    return batch_index % world


def run_rank(engine, cfg, batches, rank, world, qc_sample, paired=True):
    """Process the batches owned by `rank` (list of (batch_index, Batch) for ALL batches, in file order).
    Returns ({batch_index: result records}, statistics in capi.collect_stats form)."""
    engine.set_config(cfg)
    engine.reset_stats()
    results = {}
    for bi, batch in batches:
        if owner(bi, world) != rank:
            continue
        engine.upload(0, batch)
        engine.run(0)
        # post-filter QC while TOTAL_READS < qc_sample (preprocesser.py:624): global 0-based index < qc_sample - 1
        n_qc = batch.n if qc_sample <= 0 else max(0, min(batch.n, qc_sample - 1 - batch.first_index))
        if n_qc > 0:
            engine.qc_stat(0, capi.QC_R1_POST, 0, 0, n_qc, 1)
            if paired:
                engine.qc_stat(0, capi.QC_R2_POST, 1, 0, n_qc, 1)
        results[bi] = engine.fetch_results(0)
    return results, capi.collect_stats(engine, paired)


def gather_to_root(obj, rank, world):
    """host-side merge transport: no tensor collective, just pickled objects to rank 0"""
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(obj, bucket, dst=0)
    return bucket


def _worker(rank, world, port, qc_sample, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    res, stats = run_rank(oracle.OracleEngine(), _cfg(), _batches(), rank, world, qc_sample)
    parts = gather_to_root((res, stats), rank, world)
    if rank == 0:
        merged = capi.merge_stats([p[1] for p in parts])
        allres = {}
        for p in parts:
            allres.update(p[0])
        ret["merged"] = merged
        ret["results"] = np.concatenate([allres[k] for k in sorted(allres)])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one():
    qc_sample = 1800
    from oracle import oracle
    single_res, single = run_rank(oracle.OracleEngine(), _cfg(), _batches(), 0, 1, qc_sample)
    single_results = np.concatenate([single_res[k] for k in sorted(single_res)])

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, qc_sample, ret), nprocs=2, join=True)
    merged, results = ret["merged"], ret["results"]

    assert results.tobytes() == single_results.tobytes()
    assert merged["counters"].tolist() == single["counters"].tolist()
    assert merged["ovl"].tolist() == single["ovl"].tolist() and merged["dist"].tolist() == single["dist"].tolist()
    for w in (0, 1, 2, 3):
        assert np.array_equal(merged["qc"][w], single["qc"][w])
        km, ks = merged["kmers"][w], single["kmers"][w]

        def by_key(t):
            return {int(k): (int(c), int(o)) for k, c, o in zip(*[a.tolist() for a in t])}
        # the same dictionary: counts add up exactly, every k-mer keeps the earliest (global) first-seen key
        assert by_key(km) == by_key(ks)
        # ... so ties are ordered exactly like the sequential run
        assert capi.top_kmers(km, 8, 200) == capi.top_kmers(ks, 8, 200)
    assert int(merged["counters"][capi.C_TOTAL_READS]) == 3000


def test_owner_round_robin():
    assert [owner(i, 4) for i in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
