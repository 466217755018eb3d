#!/usr/bin/env python3
"""bench.py — Mreads/s of the AfterQC hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (aqc_run: filter / trim / overlap / correction over every pair of
the batch, plus the post-filter QC accumulation over the first qc_sample-1 records exactly as
seqFilter.run issues it for a file's first batch) over one batch of synthetic 2x150 bp pairs that is
already resident in HBM when the timed region starts.  Workload = BASELINE.json configs[2]
("10M synthetic paired-end 2x150 bp reads ... full overlap-detect + base-correction, 1xMI355X",
SURVEY.md §8d config 3, seed 1003): 5 M pairs = 10 M reads per GPU; with N GPUs every rank holds its own
5 M pairs (independent shards, no collective on the data path -> weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HIP-event kernel time,
algorithmic bytes 4L+56 per pair) and, at N=1, `cpu_baseline` (the oracle = scalar C port of the
reference loop, timed on one host core over a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
L = 150
BYTES_PER_PAIR = 4 * L + 56    # 2 seq + 2 qual + 2x12 B descriptors + 32 B result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=5_000_000, help="pairs per GPU (default: config 3 = 10 M reads)")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--qc-sample", type=int, default=200_000)
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config5"],
                    help="config3 (default, the metric's workload): PE 2x150; config2: SE 1x150 filter+trim only; "
                         "config5-like: PE 2x250 (no barcode/gzip) — the non-default ones are for DESIGN.md numbers")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    from afterqc_amd import capi, synth

    # ---- workload: seed 1003 (+rank: independent shards).  Generated with forked numpy workers, so it
    # happens BEFORE torch / the HIP runtime are loaded into this process.
    t_gen = time.time()
    nworkers = max(1, (os.cpu_count() or 8) // max(1, world))
    paired = args.workload != "config2"
    RL = 250 if args.workload == "config5" else L
    if paired:
        d = synth.make_pairs(args.pairs, RL, seed=(1003 if RL == L else 1005) + rank, workers=nworkers)
        if args.workload == "config5":
            d = synth.add_barcodes(d, 1005 + 7 + rank)      # 17-base barcode + verify prefix on both mates
        batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    else:
        d = synth.make_single(2 * args.pairs, RL, seed=1002 + rank, workers=nworkers)
        batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"])
    t_gen = time.time() - t_gen

    import torch  # plumbing only (barrier / max-over-ranks); loaded before libafterqc_hip.so so both share one HIP runtime
    dist = None
    # AQC_BENCH_SHARE_GPU=1 (debug only): all ranks on GPU 0 with the gloo backend, to exercise the multi-rank code
    # path on a one-GPU box.  The driver's runs use one GPU per rank over RCCL.
    share_gpu = os.environ.get("AQC_BENCH_SHARE_GPU") == "1"
    device_index = 0 if (world == 1 or share_gpu) else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if share_gpu:
            dist.init_process_group(backend="gloo", init_method="env://")
        else:
            dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", device_index))

    cfg = capi.Config()
    cfg.paired = 1 if paired else 0
    cfg.trim_front = cfg.trim_tail = cfg.trim_front2 = cfg.trim_tail2 = 0 if paired else 5   # config 3: -f 0 -t 0; config 2: -f 5 -t 5
    cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
    cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
    cfg.barcode_length = 12
    cfg.set_verify("CAGTA")
    cfg.qc_kmer = 8
    if args.workload == "config5":
        cfg.barcode = 1

    eng = capi.Engine(device_index, 1)
    eng.set_config(cfg)
    eng.reset_stats()
    t_up = time.perf_counter()
    eng.upload(0, batch)          # inputs resident in HBM before the timed region
    eng.sync(0)
    t_up = time.perf_counter() - t_up   # host -> HBM incl. canonicalisation; reported, never part of `value`
    n_qc = max(0, min(batch.n, args.qc_sample - 1))

    def step():
        eng.run(0)
        if n_qc:
            eng.qc_stat(0, capi.QC_R1_POST, 0, 0, n_qc, 1)
            if paired:
                eng.qc_stat(0, capi.QC_R2_POST, 1, 0, n_qc, 1)

    def barrier():
        eng.sync(0)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.timing_reset(0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync(0)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms, klaunch = eng.timing_mean(0)
    counters = eng.counters()

    ms_per_step = 1000.0 * elapsed / max(1, args.steps)
    reads_total = 2 * args.pairs * world
    records = batch.n
    bytes_per_record = (4 * RL + 56) if paired else (2 * RL + 20)
    value = reads_total / (elapsed / max(1, args.steps)) / 1e6

    k_ms = float(kms[capi.K_FILTER_OVERLAP])
    achieved = records * bytes_per_record / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("pairs") == args.pairs and args.workload == "config3":
                traffic = tj.get("bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "Mreads/s (paired 2x150 bp) end-to-end good/bad split",
        "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": ("config3: %d synthetic PE 2x150 bp pairs per GPU (%.1f M reads), seed 1003+rank, overlap ~N(30,8), "
                                "3%% adapter read-through, defaults with -f 0 -t 0, qc_sample %d; inputs resident in HBM"
                                % (args.pairs, 2 * args.pairs / 1e6, args.qc_sample)) if args.workload == "config3" else
                               ("%s: %d records of length %d per GPU, qc_sample %d; inputs resident in HBM" % (args.workload, records, RL, args.qc_sample)),
                   "pairs_per_gpu": args.pairs, "read_len": RL, "parallelism": "independent shards x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "filter_overlap_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "kernel_ms": round(k_ms, 4), "launches": int(klaunch[capi.K_FILTER_OVERLAP]),
                     "algorithmic_bytes_per_launch": records * bytes_per_record,
                     "qc_stat_kernel_ms": round(float(kms[capi.K_QC_STAT]), 4)},
        "good_reads_frac": round(float(counters[capi.C_GOOD_READS]) / max(1, float(counters[capi.C_TOTAL_READS])), 5),
        "gen_s": round(t_gen, 1),
        "upload_s": round(t_up, 3),
        "pcie_inclusive_mreads_s": round(2 * args.pairs / (t_up + elapsed / max(1, args.steps)) / 1e6, 1),
    }

    # ---- CPU baseline (rank 0, N=1 only): the oracle, 1 core, bounded sample of the same workload
    if rank == 0 and world == 1 and args.cpu_sample > 0 and paired:
        from oracle import oracle
        m = min(args.cpu_sample, args.pairs)
        sub = capi.Batch.from_matrices(d["seq1"][:m], d["qual1"][:m], d["len1"][:m], d["seq2"][:m], d["qual2"][:m], d["len2"][:m])
        oe = oracle.OracleEngine()
        oe.set_config(cfg)
        oe.upload(0, sub)
        tc = time.perf_counter()
        oe.run(0)
        tc = time.perf_counter() - tc
        # cross-check while we are here: same verdicts as the GPU for the sample
        same = bool(np.array_equal(oe.fetch_results(0), eng.fetch_results(0)[:m]))
        out["cpu_baseline"] = {"value": round(2 * m / tc / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                               "sample": "first %d pairs of the same batch, oracle/aqc_oracle.c (scalar C restatement of "
                                         "preprocesser.py:411-631), filter+overlap+correction only, %.1f s" % (m, tc),
                               "matches_gpu": same, "host_cpus": os.cpu_count()}
        # the same port on every host core (independent slices, one oracle context per thread; the C call releases the
        # GIL): SURVEY.md §8d asks for the 1-core and the all-cores figure side by side
        try:
            from concurrent.futures import ThreadPoolExecutor
            T = max(1, min(os.cpu_count() or 1, 64))
            per = max(1000, min(100_000, args.pairs // T))
            engines = []
            for k in range(T):
                lo, hi = k * per, (k + 1) * per
                e = oracle.OracleEngine()
                e.set_config(cfg)
                e.upload(0, capi.Batch.from_matrices(d["seq1"][lo:hi], d["qual1"][lo:hi], d["len1"][lo:hi],
                                                     d["seq2"][lo:hi], d["qual2"][lo:hi], d["len2"][lo:hi]))
                engines.append(e)
            ta = time.perf_counter()
            with ThreadPoolExecutor(max_workers=T) as ex:
                list(ex.map(lambda e: e.run(0), engines))
            ta = time.perf_counter() - ta
            out["cpu_baseline"]["all_cores"] = {"value": round(2 * per * T / ta / 1e6, 3), "unit": "Mreads/s", "cores": T,
                                                "sample": "%d threads x %d pairs, %.2f s" % (T, per, ta)}
        except Exception as e:      # the 1-core figure above is the contract; this one is extra
            out["cpu_baseline"]["all_cores"] = {"error": str(e)}
        # and the reference's own kind of speed: the same loop as pure Python (strings, per-base loops), 1 core —
        # the stand-in for "CPython running after.py", which cannot travel to this box
        try:
            from oracle import pyloop
            mp = min(20_000, m)
            tp = time.perf_counter()
            py = pyloop.run_batch(sub, cfg, 0, mp)
            tp = time.perf_counter() - tp
            gflags = eng.fetch_results(0)[:mp]["flag"]
            out["cpu_baseline"]["cpython_standin"] = {
                "value": round(2 * mp / tp / 1e6, 5), "unit": "Mreads/s", "cores": 1,
                "sample": "first %d pairs, oracle/pyloop.py (pure-Python restatement of the reference loop), %.1f s" % (mp, tp),
                "matches_gpu": bool(all(int(p["flag"]) == int(f) for p, f in zip(py, gflags)))}
        except Exception as e:
            out["cpu_baseline"]["cpython_standin"] = {"error": str(e)}
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
