"""Full-size properties (-m gpu): BASELINE.json's config-3 batch (5 M pairs, 2x150 bp) through the C ABI.

The oracle needs ~15 s per 5 M pairs on one core, so at this size parity is argued through size-independent
properties of the path plus an oracle check of a 1 M-pair window:
  * batch-split invariance: the verdicts, counters, histograms and QC accumulators of one 5 M-pair batch equal
    those of the same pairs fed as five 1 M-pair batches (first_index carries the global read index);
  * idempotence: a second aqc_run over the same slot reproduces every verdict byte and exactly doubles the counters;
  * conservation: every record gets exactly one verdict flag and the flag census equals the counters;
  * the multi-rank bench path (two ranks, one shared GPU, gloo) prints one well-formed JSON line.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from afterqc_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FULL = 5_000_000
QC_SAMPLE = 200_000


def config3():
    cfg = capi.Config()
    cfg.paired = 1
    cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
    cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
    cfg.barcode_length = 12
    cfg.set_verify("CAGTA")
    cfg.qc_kmer = 8
    return cfg


@pytest.fixture(scope="module")
def full():
    return synth.make_pairs(N_FULL, 150, seed=1003, workers=max(1, (os.cpu_count() or 8) // 2))


def sub_batch(d, lo, hi):
    return capi.Batch.from_matrices(d["seq1"][lo:hi], d["qual1"][lo:hi], d["len1"][lo:hi],
                                    d["seq2"][lo:hi], d["qual2"][lo:hi], d["len2"][lo:hi], first_index=lo)


def snapshot(eng):
    keys, counts, order = eng.kmers(capi.QC_R1_POST)
    idx = np.argsort(order, kind="stable")
    return dict(counters=eng.counters().copy(), hist=[h.copy() for h in eng.histograms()],
                qc=[eng.qc(w).copy() for w in (capi.QC_R1_POST, capi.QC_R2_POST)],
                kmers=(keys[idx].copy(), counts[idx].copy()))


def run_batches(eng, d, bounds, sample=QC_SAMPLE):
    """what seqFilter.run issues: per batch the hot path, then the post-filter sampling while TOTAL_READS < qc_sample"""
    eng.set_config(config3())
    eng.reset_stats()
    res = []
    for lo, hi in bounds:
        eng.upload(0, sub_batch(d, lo, hi))
        eng.run(0)
        n_qc = max(0, min(hi, sample - 1) - lo)
        if n_qc:
            eng.qc_stat(0, capi.QC_R1_POST, 0, 0, n_qc, 1)
            eng.qc_stat(0, capi.QC_R2_POST, 1, 0, n_qc, 1)
        res.append(eng.fetch_results(0))
    return np.concatenate(res), snapshot(eng)


def same_snapshot(a, b):
    assert a["counters"].tolist() == b["counters"].tolist()
    for x, y in zip(a["hist"], b["hist"]):
        assert np.array_equal(x, y)
    for x, y in zip(a["qc"], b["qc"]):
        assert np.array_equal(x, y)
    assert np.array_equal(a["kmers"][0], b["kmers"][0]) and np.array_equal(a["kmers"][1], b["kmers"][1])


def test_full_size_properties(gpu_engine, full):
    eng = gpu_engine
    whole, snap_whole = run_batches(eng, full, [(0, N_FULL)])
    assert len(whole) == N_FULL

    # conservation: one flag per record, census == counters
    flags = np.bincount(whole["flag"], minlength=capi.N_FLAGS)
    c = snap_whole["counters"]
    assert int(flags.sum()) == N_FULL
    assert int(c[capi.C_TOTAL_READS]) == N_FULL          # the reference counts loop iterations (records)
    assert int(c[capi.C_GOOD_READS]) == int(flags[capi.GOOD])
    assert [int(v) for v in c[capi.C_FLAG0:capi.C_FLAG0 + capi.N_FLAGS]] == [int(v) for v in flags]

    # idempotence: same slot again -> identical verdict bytes, counters exactly doubled
    eng.run(0)
    again = eng.fetch_results(0)
    assert np.array_equal(whole.view(np.uint8), again.view(np.uint8))
    assert (eng.counters() == 2 * c).all()

    # batch-split invariance (five 1 M-pair batches, ragged last split)
    bounds = [(0, 1_000_000), (1_000_000, 2_300_000), (2_300_000, 2_300_001), (2_300_001, 4_000_000), (4_000_000, N_FULL)]
    parts, snap_parts = run_batches(eng, full, bounds)
    assert np.array_equal(whole.view(np.uint8), parts.view(np.uint8))
    same_snapshot(snap_whole, snap_parts)

    # a 1 M-pair window in the middle against the oracle (verdict bytes + counters)
    from oracle import oracle
    lo, hi = 2_000_000, 3_000_000
    oe = oracle.OracleEngine()
    oe.set_config(config3())
    oe.reset_stats()
    oe.upload(0, sub_batch(full, lo, hi))
    oe.run(0)
    assert np.array_equal(oe.fetch_results(0).view(np.uint8), whole[lo:hi].view(np.uint8))


def test_sample_split_across_batches(gpu_engine, full):
    """the qc_sample boundary falling inside a later batch: sampling state (time keys, k-mer order) must not depend
    on where the batch boundaries are"""
    eng = gpu_engine
    n = 400_000
    _, one = run_batches(eng, full, [(0, n)])
    _, many = run_batches(eng, full, [(0, 70_000), (70_000, 150_001), (150_001, 260_000), (260_000, n)])
    same_snapshot(one, many)


def test_bench_two_ranks_shared_gpu(tmp_path):
    """bench.py under torch.distributed.run with 2 ranks (gloo, both on GPU 0): one JSON line, n_gpus == 2,
    aggregate reads of both ranks"""
    env = dict(os.environ, AQC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--pairs", "500000", "--cpu-sample", "0"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    assert "cpu_baseline" not in out          # rank 0 at N=1 only
    assert abs(out["value"] - 2 * 2 * 500000 / (out["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * out["value"]
    # BOTH shapes in one line: `value` = one input dealt over the two ranks' devices (two output files: flat by design), and
    # multi_input_file_to_file = one input PER rank, every rank its own pipe over its own files, at once (the shape that scales)
    assert "one input" in out["config"]["parallelism"] or "ONE input" in out["config"]["parallelism"], out["config"]
    mi = out["multi_input_file_to_file"]
    assert mi and mi["inputs"] == 2 and mi["output_files"] == 8 and mi["mreads_s"] > 0, mi


def test_config4_size_soak_one_input_of_100M_reads(tmp_path):
    """BASELINE config 4 at its stated size on one GPU (tools/soak_config4.py: ONE input of 10 x the 5 M-pair block = 100 M reads,
    two 17 GB files, over two contexts on GPU 0): every output file equals the block's output x 10 byte for byte, counters and
    histograms are 10 x the block's, the post-filter QC rows the block's, record indices beyond 2^32 give the same bytes.  The
    inputs live in /dev/shm (35 GB would not fit beside 34 GB of outputs on the boxes' 79 GB disks) and are clean when the run
    starts; the run's rate is printed, and must not fall back to the write-back-throttled 20 Mreads/s of round 4's soak."""
    import shutil
    in_dir = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 45e9 else None
    copies = 10 if in_dir and shutil.disk_usage(str(tmp_path)).free > 45e9 else 3
    cmd = [sys.executable, os.path.join(ROOT, "tools", "soak_config4.py"), "--copies", str(copies), "--devices", "0,0", "--dir", str(tmp_path)]
    if in_dir:
        cmd += ["--in-dir", in_dir]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    log = json.loads(p.stdout[p.stdout.index("{"):])
    assert log["ok"] and log["copies"] == copies and log["reads"] == 2 * 5_000_000 * copies
    assert log["counters_equal_copies_x_block"] and log["histograms_equal_copies_x_block"] and log["post_filter_qc_rows_equal_block"]
    assert all(v["equals_block_output_x_copies"] for v in log["outputs"].values())
    assert log["indices_beyond_2_32"]["identical_outputs_and_counters"]
    print("soak:", json.dumps(log["run"]))
    if copies == 10:
        assert log["largest_output_gib"] > 15 and log["run"]["mreads_s"] > 28, log["run"]
