"""Multi-GPU sharding of the hot path.

Every record is independent; the only cross-record state is additive (counters, histograms, QC
rows) or a min/sum merge (k-mer dictionary: counts add, insertion time keys take the minimum —
the device stamps them with the *global* read index, so the merged dictionary orders ties
exactly like a sequential run).  Hence: fixed-size batches are dealt round-robin to the ranks,
each batch carries its `first_index`, there is NO collective on the data path, and the per-rank
statistics are summed on the host (rank 0) with `gather_object`.  Outputs are stitched in batch
order.
"""
import numpy as np

from . import capi


def owner(batch_index, world):
    """rank that processes batch `batch_index`"""
    return batch_index % world


def collect(engine, paired=True):
    """Pull every statistic of one engine into plain numpy / dict form."""
    whichs = (0, 1, 2, 3) if paired else (capi.QC_R1_PRE, capi.QC_R1_POST)
    ovl, dist = engine.histograms()
    km = {}
    for w in whichs:
        keys, counts, order = engine.kmers(w)
        km[w] = {int(k): (int(c), int(o)) for k, c, o in zip(keys.tolist(), counts.tolist(), order.tolist())}
    return {"counters": engine.counters(), "ovl": ovl, "dist": dist, "qc": {w: engine.qc(w) for w in whichs}, "kmers": km}


def merge(parts):
    """Sum the additive statistics of several ranks; merge k-mer dictionaries by (count sum, min time key)."""
    out = {"counters": sum(p["counters"] for p in parts), "ovl": sum(p["ovl"] for p in parts),
           "dist": sum(p["dist"] for p in parts), "qc": {}, "kmers": {}}
    for w in parts[0]["qc"]:
        out["qc"][w] = sum(p["qc"][w] for p in parts)
        d = {}
        for p in parts:
            for k, (c, o) in p["kmers"][w].items():
                if k in d:
                    d[k] = (d[k][0] + c, min(d[k][1], o))
                else:
                    d[k] = (c, o)
        out["kmers"][w] = d
    return out


def top_kmers(kmer_dict, kmer_len, top=10):
    """sortKmer (qualitycontrol.py:155-156) on a merged dictionary: count descending, insertion order for ties"""
    items = sorted(kmer_dict.items(), key=lambda kv: (-kv[1][0], kv[1][1]))[:top]
    return [[int(k).to_bytes(8, "little")[:kmer_len].decode("latin-1"), c] for k, (c, o) in items]


def run_rank(engine, cfg, batches, rank, world, qc_sample, paired=True):
    """Process the batches owned by `rank` (list of (batch_index, Batch) for ALL batches, in file order).
    Returns ({batch_index: result records}, statistics)."""
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
    return results, collect(engine, paired)


def gather_to_root(obj, dist, rank, world):
    """host-side merge transport: no tensor collective, just pickled objects to rank 0"""
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(obj, bucket, dst=0)
    return bucket
