#!/usr/bin/env python3
"""bench.py — Mreads/s of the AfterQC hot path on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2] (seed 1003): 5 M synthetic 2x150 bp pairs = 10 M reads per
GPU, as RAW FASTQ TEXT (two mate files, 1.7 GB each).

`value` (the metric: Mreads/s END TO END, good / bad split) — one STEP = the product's whole-input pipe (aqc_pipe_run, what
`after.py -1 R1.fq -2 R2.fq` runs) from the two FASTQ FILES to the four good / bad FILES: parallel pread into page-locked
rings -> H2D -> record framing -> verdicts -> post-filter QC sampling -> good / bad text -> D2H -> one sequential writer
per output file.  ONE input whatever N is: with N GPUs the chunks of an N x 5 M-pair input are dealt round robin over N
contexts by ONE process (rank 0 drives all N devices — SURVEY.md §8e, BASELINE config 4: no collective; the other ranks
of a torchrun launch only take part in the barriers and in the device-step measurement below).  `--contexts C` puts C
contexts on every device (for a one-GPU box: the N-context code path on one device).

Other numbers on the same JSON line (never `value`):
  device_step_mreads_s         the same work with the text RESIDENT IN HBM: aqc_reframe -> aqc_run -> aqc_qc_stat ->
                               aqc_format per chunk, no PCIe, no files (each rank on its own GPU; what rounds 1-2 printed as
                               `value`).  `roofline` is for its dominant kernel (fast_filter_overlap_kernel): HIP-event time
                               on the slot's stream, algorithmic bytes 4L+56 per pair.
  pinned_to_pinned_mreads_s    the pipe fed from / fetched into page-locked host memory (PCIe inclusive, no files)
  file_to_file_gz              the pipe .gz -> .gz: ONE-member `gzip -2` inputs decoded by the host pool AND the GPU (groups of sections,
                               a lane per deflate block: aqc_gunzip_dev.hpp; `gunzip_text_share_from_device`), .gz members built on the device (--gz-runs; by default three runs when the input is the 1-GPU one)
  file_to_gz                   plain FASTQ in, .gz out (the outputs a third of the size: the file writers are not the bound)
  multi_input_file_to_file     K (--inputs, default 2) independent inputs through K pipes at once on the same GPU: the reference's own
                               parallelism is one seqFilter per input file (after.py:168-171), and K inputs write 4 K files
`cpu_baseline` (N=1): the oracle (scalar C port of the reference loop) on one core over a bounded sample, checked against the
GPU's verdicts; plus the same port on every host core and a pure-Python stand-in.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
L = 150


def hbm_traffic(args, n_rec, live):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
    passes, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KB and tallies the 128-byte requests of a streaming
    read at 64 bytes on gfx950, so it is doubled; WRITE_SIZE in KB as reported).  Counters cannot be read from inside the process
    that is being measured: with `live` two short child runs of this script (`--device-only`, the same workload) are made under
    rocprofv3 on this box; otherwise, or when that fails, the committed measurement of the same workload is quoted
    (profiles/hbm_traffic.json).  `traffic_source` in the JSON line says which."""
    import glob
    import subprocess
    rp = shutil.which("rocprofv3")
    if live and rp and not os.environ.get("AQC_BENCH_CHILD"):
        tmp = tempfile.mkdtemp(prefix="aqc_pmc_")
        try:
            kb = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                cmd = [rp, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--device-only",
                       "--device-steps", "3", "--cpu-sample", "0", "--no-fused-step", "--workload", args.workload, "--pairs", str(int(n_rec))]
                subprocess.run(cmd, cwd=tmp, env=dict(os.environ, AQC_BENCH_CHILD="1", TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=150, check=True)
                tot, cnt = 0.0, 0
                import csv
                for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(fn) as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] == counter and "fast_filter_overlap_kernel" in row["Kernel_Name"]:
                                tot += float(row["Counter_Value"])
                                cnt += 1
                if not cnt:
                    raise RuntimeError("no dispatch of the kernel in the counter file")
                kb[counter] = tot / cnt
            return (int(kb["FETCH_SIZE"] * 1024 * 2 + kb["WRITE_SIZE"] * 1024),
                    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over two child runs of this script on this box (--device-only --device-steps 3): "
                    "FETCH_SIZE %.0f KB x 2 + WRITE_SIZE %.0f KB per launch" % (kb["FETCH_SIZE"], kb["WRITE_SIZE"]))
        except Exception as e:          # noqa: BLE001 — profiler missing / refused: quote the committed measurement instead
            sys.stderr.write("bench: live PMC pass failed (%s: %s); quoting profiles/hbm_traffic.json\n" % (type(e).__name__, e))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)
        if t.get("workload", "config3") == args.workload and int(t["pairs"]) == int(n_rec):
            return int(t["bytes_per_launch"]), "profiles/hbm_traffic.json (%s)" % t.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=5_000_000, help="pairs per GPU (default: config 3 = 10 M reads)")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--qc-sample", type=int, default=200_000)
    ap.add_argument("--chunk-records", type=int, default=1 << 17, help="records per chunk of the pipe")
    ap.add_argument("--pipe-runs", type=int, default=3, help="timed runs of the pinned->pinned pipe (0 = skip)")
    ap.add_argument("--device-steps", type=int, default=10, help="timed steps of the HBM-resident device pipeline (0 = skip)")
    ap.add_argument("--gz-runs", type=int, default=-1, help="timed .gz -> .gz runs of the pipe (0 = skip; default: 3 for the 1-GPU input, where making "
                    "the inputs with gzip -2 takes ~20 s, else 0)")
    ap.add_argument("--no-fused-step", action="store_true", help="skip the AQC_FUSED=1 variant of the device step (a second context on the same GPU)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc child runs (quote profiles/hbm_traffic.json)")
    ap.add_argument("--device-only", action="store_true", help="profiling runs (rocprofv3): only the HBM-resident device step, no pipe "
                    "runs; `value` is then the device step and says so")
    ap.add_argument("--text-step-only", action="store_true", help="with --device-only: skip the aqc_format_spans variant of the device step (profiles of the text step alone)")
    ap.add_argument("--spans-step-only", action="store_true", help="with --device-only: run only the aqc_format_spans variant of the device step in the timed region (profiles)")
    ap.add_argument("--slots", type=int, default=3, help="slots (chunks in flight) per context of the pipe")
    ap.add_argument("--inputs", type=int, default=2, help="K > 1: also run K independent file pairs through K pipes at once on the same GPU(s) — the "
                    "reference's own fan-out, one seqFilter per input (after.py:168-171) — reported as multi_input_file_to_file (never `value`); 0 / 1 = skip")
    ap.add_argument("--big-copies", type=int, default=10, help="N = 1: also run ONE input of this many copies of the workload (10 = config 4's stated size, 100 M reads) "
                    "file -> file once, reported as file_to_file_100M (never `value`); inputs go to /dev/shm when it has the room; 0 / 1 = skip")
    ap.add_argument("--contexts", type=int, default=1, help="contexts per device for the one-input pipe runs")
    ap.add_argument("--devices", default="", help="explicit device list for the one-input pipe runs, e.g. 0,0,0,0 (overrides --gpus / --contexts)")
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config5"],
                    help="config3 (default, the metric's workload): PE 2x150; config2: SE 1x150 filter+trim only; "
                         "config5: PE 2x250 + 17-base barcode / verify prefix, barcode mode on")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    from afterqc_amd import capi, synth

    # ---- workload (generated with forked numpy workers BEFORE torch / the HIP runtime are loaded into this process)
    t_gen = time.time()
    nworkers = max(1, (os.cpu_count() or 8) // max(1, world))
    paired = args.workload != "config2"
    RL = 250 if args.workload == "config5" else L
    if paired:
        d = synth.make_pairs(args.pairs, RL, seed=(1003 if RL == L else 1005) + rank, workers=nworkers)
        if args.workload == "config5":
            d = synth.add_barcodes(d, 1005 + 7 + rank)      # 17-base barcode + verify prefix on both mates
    else:
        d = synth.make_single(2 * args.pairs, RL, seed=1002 + rank, workers=nworkers)
    n_rec = len(d["len1"])
    W = d["seq1"].shape[1]
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

    eng = capi.Engine(device_index, max(3, args.slots))
    eng.set_config(cfg)
    eng.reset_stats()

    # ---- the input as raw FASTQ text in page-locked host memory (what a reader thread would have produced)
    t_txt = time.time()
    w = synth.fixed_record_width(W)
    texts = []
    for mate in ((1, 2) if paired else (1,)):
        hb = eng.host_buffer(n_rec * w + 4096)
        _, nbytes = synth.render_fastq_fixed(d["seq%d" % mate], d["qual%d" % mate], mate, out=hb.array, index0=rank * n_rec)
        texts.append((hb, nbytes))
    t_txt = time.time() - t_txt
    K = args.chunk_records
    first_index = rank * n_rec
    reads_per_gpu = n_rec * (2 if paired else 1)
    text_in = sum(t[1] for t in texts)

    def barrier():
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ================= A. the device step: text resident in HBM -> good / bad text in HBM (every rank, its own GPU) =========
    # (a chunk is < 2 GiB of text per mate: 5 M records of 347 bytes fit one chunk, longer inputs are spread over the slots)
    per_slot = min(n_rec, int(1.9e9 // w))
    n_res = (n_rec + per_slot - 1) // per_slot
    assert n_res <= eng.n_slots, "input too large to keep resident in %d slots" % eng.n_slots
    t_up = time.perf_counter()
    res_n = []
    for sl in range(n_res):
        lo, hi = sl * per_slot, min(n_rec, (sl + 1) * per_slot)
        views = [(t[0].array[lo * w:], (hi - lo) * w) for t in texts]
        if paired:
            info = eng.frame(sl, views[0][0], views[0][1], True, views[1][0], views[1][1], True, first_index=first_index + lo)
        else:
            info = eng.frame(sl, views[0][0], views[0][1], True, first_index=first_index + lo)
        assert int(info.n) == hi - lo, (int(info.n), hi - lo)
        res_n.append((lo, hi))
    for sl in range(n_res):
        eng.sync(sl)
    t_up = time.perf_counter() - t_up          # host -> HBM + first framing (reported)

    def step(spans=False):
        total = [0] * 6
        for sl, (lo, hi) in enumerate(res_n):
            eng.reframe(sl)
            eng.run(sl)
            n_qc = (hi - lo) if args.qc_sample <= 0 else max(0, min(hi - lo, args.qc_sample - 1 - (first_index + lo)))
            if n_qc > 0:
                eng.qc_stat(sl, capi.QC_R1_POST, 0, 0, n_qc, 1)
                if paired:
                    eng.qc_stat(sl, capi.QC_R2_POST, 1, 0, n_qc, 1)
            # (returns once the sizes are known; the writer kernels are still queued)
            if spans:
                # what aqc_pipe_run does for plain-text outputs: the good records that go out as their own bytes are not copied
                # (they are written from the host's input buffer), stream 0 holds only the rebuilt ones + the event list
                sz, _ = eng.format_spans(sl, hi - lo, False)
            else:
                sz = eng.format(sl, hi - lo, False)
            total = [a + b for a, b in zip(total, sz)]
        return total

    def sync_all():
        for sl in range(n_res):
            eng.sync(sl)

    dsteps = max(1, args.device_steps)
    for _ in range(min(3, max(1, args.warmup))):
        sizes = step()
    sync_all()
    barrier()
    for sl in range(n_res):
        eng.timing_reset(sl)
    t0 = time.perf_counter()
    for _ in range(dsteps if not args.spans_step_only else 0):
        sizes = step()
    sync_all()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dev_elapsed = max_over_ranks(time.perf_counter() - t0)
    # ... and the same step the way the pipe runs it for plain-text outputs (aqc_format_spans)
    spans_sizes = step(True)
    sync_all()
    barrier()
    t0 = time.perf_counter()
    for _ in range(dsteps if not args.text_step_only else 0):
        spans_sizes = step(True)
    sync_all()
    spans_elapsed = max_over_ranks(time.perf_counter() - t0)
    # ... and the text step on a context created with AQC_FUSED=1 (opt-in): aqc_run's verdict kernel places every record
    # and copies the whole good ones itself, aqc_format rebuilds the rest.  Same bytes; reported beside the default step.
    fused = None
    if paired and args.workload == "config3" and not args.spans_step_only and not args.no_fused_step:
        old_f = os.environ.get("AQC_FUSED")
        os.environ["AQC_FUSED"] = "1"
        try:
            eng_f = capi.Engine(device_index, max(1, n_res))
        finally:
            if old_f is None:
                os.environ.pop("AQC_FUSED", None)
            else:
                os.environ["AQC_FUSED"] = old_f
        eng_f.set_config(cfg)
        eng_f.reset_stats()
        for sl, (lo, hi) in enumerate(res_n):
            views = [(t[0].array[lo * w:], (hi - lo) * w) for t in texts]
            eng_f.frame(sl, views[0][0], views[0][1], True, views[1][0], views[1][1], True, first_index=first_index + lo)

        def step_fused():
            total, taken = [0] * 6, True
            for sl, (lo, hi) in enumerate(res_n):
                eng_f.reframe(sl)
                eng_f.run(sl)
                n_qc = (hi - lo) if args.qc_sample <= 0 else max(0, min(hi - lo, args.qc_sample - 1 - (first_index + lo)))
                if n_qc > 0:
                    eng_f.qc_stat(sl, capi.QC_R1_POST, 0, 0, n_qc, 1)
                    eng_f.qc_stat(sl, capi.QC_R2_POST, 1, 0, n_qc, 1)
                sz = eng_f.format(sl, hi - lo, False)
                taken = taken and eng_f.format_fused(sl)
                total = [a + b for a, b in zip(total, sz)]
            return total, taken

        for _ in range(2):
            fused_sizes, fused_taken = step_fused()
        for sl in range(n_res):
            eng_f.sync(sl)
        barrier()
        t0 = time.perf_counter()
        for _ in range(dsteps):
            fused_sizes, fused_taken = step_fused()
        for sl in range(n_res):
            eng_f.sync(sl)
        fused_elapsed = max_over_ranks(time.perf_counter() - t0)
        fused = {"ms_per_step": round(1000.0 * fused_elapsed / dsteps, 4), "mreads_s": round(reads_per_gpu * world / max(fused_elapsed / dsteps, 1e-9) / 1e6, 2),
                 "placement_taken": bool(fused_taken), "same_sizes_as_device_step": [int(x) for x in fused_sizes] == [int(x) for x in sizes] if not args.spans_step_only else None,
                 "what": "the text step on a context created with AQC_FUSED=1: fast_filter_overlap_kernel<10,true,12,false,true> gives every record its place "
                         "(in-batch scans + decoupled look-back over the batches) and copies the good records that go out as their own bytes; aqc_format "
                         "patches / rebuilds the rest (no sizing passes, no whole-record copy kernel)"}
        eng_f.close()
    kms, klaunch = eng.timing_mean(0)
    for sl in range(1, n_res):
        k2, l2 = eng.timing_mean(sl)
        kms = kms + k2                    # per step the kernel runs once per resident chunk: times add up
    counters = eng.counters()
    res_gpu = eng.fetch_results(0) if (rank == 0 and world == 1 and args.cpu_sample > 0 and paired) else None
    if n_res > 1:
        # With several chunks resident their slots' streams overlap: the dominant kernel of one chunk shares the GPU with
        # the writer kernels of another and its event time says little about the kernel.  For the roofline line the
        # kernel is timed again alone, chunk after chunk.
        for sl in range(n_res):
            eng.timing_reset(sl)
        for _ in range(dsteps):
            for sl in range(n_res):
                eng.run(sl)
                eng.sync(sl)
        k_alone = eng.timing_mean(0)[0]
        for sl in range(1, n_res):
            k_alone = k_alone + eng.timing_mean(sl)[0]
        kms[capi.K_FILTER_OVERLAP] = k_alone[capi.K_FILTER_OVERLAP]
    dev_ms = 1000.0 * dev_elapsed / dsteps
    dev_value = reads_per_gpu * world / (dev_elapsed / dsteps) / 1e6
    bytes_per_record = (4 * RL + 56) if paired else (2 * RL + 20)
    k_ms = float(kms[capi.K_FILTER_OVERLAP])
    achieved = n_rec * bytes_per_record / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    text_out = int(sum(sizes))
    good_frac = float(counters[capi.C_GOOD_READS]) / max(1.0, float(counters[capi.C_TOTAL_READS]))

    # ================= B/C. ONE input through the whole-input pipe, driven by rank 0 over every device ========================
    if args.devices:
        dev_list = [int(x) for x in args.devices.split(",")]
    elif share_gpu:
        dev_list = [0] * (world * max(1, args.contexts))
    else:
        dev_list = [g for g in range(world) for _ in range(max(1, args.contexts))]
    n_ctx = len(dev_list)
    copies = max(1, world)                      # the one input holds `world` x the per-GPU share (weak scaling, one input)
    pinned = f2f = f2f_gz = f2gz = multi = big = None
    step_times = []
    pipe_reads = reads_per_gpu * copies
    if args.device_only:
        args.pipe_runs = 0
        args.warmup = args.steps = 0
    # One input over N GPUs needs rank 0 to SEE the N devices.  When the launcher narrows every rank's visibility to its own GPU
    # (HIP_VISIBLE_DEVICES per rank) that is impossible: then every rank runs its OWN pipe on its own GPU over its own share of
    # the input (independent batches, as north_star's shards; files R<k>.rank<r>.*), and `value` = all ranks' reads / the slowest
    # rank's time.  config.parallelism says which of the two it was.
    per_rank = False
    if world > 1 and not args.devices:
        n_visible = capi.load_library().aqc_device_count()
        seen = torch.tensor([n_visible], dtype=torch.int64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(seen, op=dist.ReduceOp.MIN)
        per_rank = (not share_gpu and int(seen.item()) < world) or os.environ.get("AQC_BENCH_PER_RANK") == "1"
    if per_rank:
        dev_list = [device_index] * max(1, args.contexts)
        n_ctx = len(dev_list)
        copies = 1
        pipe_reads = reads_per_gpu * world
    driver = rank == 0 or per_rank               # this rank drives a pipe
    if driver:
        engines = [eng]
        for g in dev_list[1:]:
            e2 = capi.Engine(g, max(3, args.slots))
            e2.set_config(cfg)
            e2.reset_stats()
            engines.append(e2)
        pipe = capi.Pipe(engines, slots=args.slots)
    else:
        engines, pipe = [], None

    def reset_all():
        for e in engines:
            e.reset_stats()

    # ---- B. page-locked memory -> page-locked memory (PCIe inclusive): this rank's share only
    if args.pipe_runs > 0 and rank == 0:
        inputs = [(t[0].array, t[1]) for t in texts]
        ts = []
        for it in range(args.pipe_runs + 1):            # the first run allocates the page-locked rings: not timed
            reset_all()
            t1 = time.perf_counter()
            pr = pipe.run(inputs, outputs=None, chunk_records=K, qc_sample=args.qc_sample)
            dt = time.perf_counter() - t1
            assert not pr.anomaly and int(pr.records) == n_rec
            if it:
                ts.append(dt)
        best = min(ts)
        pinned = {"mreads_s": round(reads_per_gpu / best / 1e6, 2), "seconds": round(best, 4), "runs": len(ts), "contexts": n_ctx,
                  "gb_s_each_way": round(text_in / best / 1e9, 2), "chunk_records": K}
        # ... and with AQC_SPANS=1 (opt-in, see aqc_pipe.cpp): the good records that go out as their own bytes are neither copied on the
        # device nor downloaded — the text goes up, only the bad / rebuilt records and the event lists come back
        old_spans = os.environ.get("AQC_SPANS")
        os.environ["AQC_SPANS"] = "1"
        try:
            ts = []
            for it in range(args.pipe_runs + 1):
                reset_all()
                t1 = time.perf_counter()
                pr = pipe.run(inputs, outputs=None, chunk_records=K, qc_sample=args.qc_sample)
                dt = time.perf_counter() - t1
                assert not pr.anomaly and int(pr.records) == n_rec
                if it:
                    ts.append(dt)
            pinned["spans_mreads_s"] = round(reads_per_gpu / min(ts) / 1e6, 2)
            pinned["spans_gb_s_up"] = round(text_in / min(ts) / 1e9, 2)
        finally:
            if old_spans is None:
                os.environ.pop("AQC_SPANS", None)
            else:
                os.environ["AQC_SPANS"] = old_spans

    # ---- C. file -> pipe -> file: THE METRIC.  W warm-up runs, then K timed runs, each bracketed by a barrier + device
    #      synchronisation on both sides; the previous run's output files are unlinked between runs (not timed: dropping
    #      3.4 GB of page cache belongs to no run).  Files live in the temp dir (page cache); AQC_BENCH_DIR picks another place.
    base = os.environ.get("AQC_BENCH_DIR") or None
    work = tempfile.mkdtemp(prefix="aqc_bench_%d_" % rank, dir=base) if (rank == 0 and not args.device_only) else None
    try:
        paths, outs = [], []
        tag = (".rank%d" % rank) if per_rank else ""
        if per_rank and not args.device_only:
            work = tempfile.mkdtemp(prefix="aqc_bench_%d_" % rank, dir=base)
        if driver and not args.device_only:
            for k, t in enumerate(texts):
                p = os.path.join(work, "R%d%s.fq" % (k + 1, tag))
                with open(p, "wb") as f:
                    for _ in range(copies):
                        f.write(memoryview(t[0].array)[:t[1]])
                paths.append(p)
            outs = [(os.path.join(work, "R%d%s.good.fq" % (k + 1, tag)), os.path.join(work, "R%d%s.bad.fq" % (k + 1, tag)), None) for k in range(len(texts))]
        last = None
        for it in range(args.warmup + args.steps):
            if driver:
                reset_all()
                for trio in outs:
                    for pth in trio:
                        if pth and os.path.exists(pth):
                            os.unlink(pth)
            barrier()
            t1 = time.perf_counter()
            if driver:
                last = pipe.run(paths, outs, chunk_records=K, qc_sample=args.qc_sample)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            dt = max_over_ranks(time.perf_counter() - t1)
            if driver:
                assert not last.anomaly and int(last.records) == n_rec * copies, (last.anomaly, int(last.records))
            if it >= args.warmup:
                step_times.append(dt)
        if rank == 0 and step_times:
            f2f = {"seconds_mean": round(sum(step_times) / len(step_times), 4), "seconds_min": round(min(step_times), 4),
                   "seconds_median": round(sorted(step_times)[len(step_times) // 2], 4),
                   "seconds_max": round(max(step_times), 4), "where": base or tempfile.gettempdir(),
                   "input_gb": round(text_in * copies / 1e9, 3), "output_gb": round(sum(int(x) for x in last.bytes_out) / 1e9, 3),
                   "contexts": n_ctx, "devices": dev_list, "slots": args.slots, "thread_seconds_last_run": last.breakdown()}
        # ---- K inputs at once (rank 0, N = 1 runs): K file pairs, K contexts on the device(s), K pipes, 4 K output files.  One input
        #      cannot be written faster than its two big output files take (~10 GB/s per file on these hosts); the
        #      reference's own parallelism is per input file, and that shape is not bound by one file's write rate.
        if args.inputs > 1 and rank == 0 and world == 1 and not args.device_only and step_times:
            import threading
            KI = args.inputs
            mi_eng, mi_pipe, mi_paths, mi_outs = [eng], [pipe], [paths], [outs]
            for k in range(1, KI):
                e2 = capi.Engine(dev_list[k % len(dev_list)], max(3, args.slots))
                e2.set_config(cfg)
                e2.reset_stats()
                mi_eng.append(e2)
                mi_pipe.append(capi.Pipe([e2], slots=args.slots))
                pk = []
                for j, src in enumerate(paths):
                    dst = os.path.join(work, "in%d_R%d.fq" % (k, j + 1))
                    shutil.copyfile(src, dst)
                    pk.append(dst)
                mi_paths.append(pk)
                mi_outs.append([(os.path.join(work, "in%d_R%d.good.fq" % (k, j + 1)), os.path.join(work, "in%d_R%d.bad.fq" % (k, j + 1)), None) for j in range(len(paths))])
            mi_times = []
            for it in range(3):
                for e in mi_eng:
                    e.reset_stats()
                for oo in mi_outs:
                    for trio in oo:
                        for pth in trio:
                            if pth and os.path.exists(pth):
                                os.unlink(pth)
                res_k = [None] * KI

                def one(k):
                    res_k[k] = mi_pipe[k].run(mi_paths[k], mi_outs[k], chunk_records=K, qc_sample=args.qc_sample)
                th = [threading.Thread(target=one, args=(k,)) for k in range(KI)]
                t1 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dt = time.perf_counter() - t1
                assert all(r is not None and not r.anomaly and int(r.records) == n_rec for r in res_k)
                if it:
                    mi_times.append(dt)
            multi = {"inputs": KI, "mreads_s": round(KI * reads_per_gpu / min(mi_times) / 1e6, 2), "seconds": round(min(mi_times), 4), "runs": len(mi_times),
                     "output_files": 2 * KI * len(paths), "what": "%d independent inputs of %.1f M reads each, one pipe + one context per input, all on device(s) %s, at once"
                     % (KI, reads_per_gpu / 1e6, sorted(set(dev_list)))}
            for k in range(1, KI):
                mi_pipe[k].close()
                mi_eng[k].close()
                for pth in mi_paths[k]:
                    os.unlink(pth)
                for trio in mi_outs[k]:
                    for pth in trio:
                        if pth and os.path.exists(pth):
                            os.unlink(pth)
        # ---- config 4 at its STATED size on one GPU: ONE input of big_copies x the workload (100 M reads: two 17 GB files in, two 16 GB
        #      good files out).  The inputs are made clean first (tmpfs, or sync()): a run that starts with 35 GB of dirty input pages
        #      in the cache hits the cgroup's dirty limit half way through its own 34 GB of output and is throttled to the disk's
        #      write-back rate (round 4's soak: 20 Mreads/s) — which says something about the bench, not about the pipe.
        if rank == 0 and world == 1 and args.big_copies > 1 and not args.device_only and step_times and args.workload == "config3":
            BC = args.big_copies
            need_out = int(1.15 * BC * text_out_estimate) if (text_out_estimate := int(sum(last.bytes_out))) else 0
            shm = "/dev/shm"
            in_dir = shm if (os.path.isdir(shm) and shutil.disk_usage(shm).free > 1.2 * BC * text_in) else work
            free_out = shutil.disk_usage(work).free - (0 if in_dir == shm else int(1.05 * BC * text_in))
            if need_out and free_out > need_out:
                big_work = tempfile.mkdtemp(prefix="aqc_bench_big_", dir=in_dir)
                try:
                    big_paths = []
                    t_w = time.perf_counter()
                    for k, t in enumerate(texts):
                        pth = os.path.join(big_work, "R%d.fq" % (k + 1))
                        with open(pth, "wb") as f:
                            for _ in range(BC):
                                f.write(memoryview(t[0].array)[:t[1]])
                        big_paths.append(pth)
                    if in_dir != shm:
                        os.sync()
                    t_w = time.perf_counter() - t_w
                    big_outs = [(os.path.join(work, "big_R%d.good.fq" % (k + 1)), os.path.join(work, "big_R%d.bad.fq" % (k + 1)), None) for k in range(len(texts))]
                    for trio in outs:                          # (the 10 M-read run's outputs: their dirty pages are not this run's business)
                        for pth in trio:
                            if pth and os.path.exists(pth):
                                os.unlink(pth)
                    os.sync()
                    reset_all()
                    t1 = time.perf_counter()
                    rb = pipe.run(big_paths, big_outs, chunk_records=K, qc_sample=args.qc_sample)
                    if torch.cuda.is_available():
                        torch.cuda.synchronize()
                    dtb = time.perf_counter() - t1
                    assert not rb.anomaly and int(rb.records) == n_rec * BC, (rb.anomaly, int(rb.records))
                    big = {"reads": reads_per_gpu * BC, "seconds": round(dtb, 3), "mreads_s": round(reads_per_gpu * BC / dtb / 1e6, 2), "input_gb": round(BC * text_in / 1e9, 2),
                           "output_gb": round(sum(rb.bytes_out) / 1e9, 2), "inputs_in": in_dir + (" (tmpfs)" if in_dir == shm else " (sync()ed)"), "outputs_in": work,
                           "make_inputs_s": round(t_w, 2), "thread_seconds": rb.breakdown(), "runs": 1,
                           "what": "ONE input of %d x the 10 M-read workload (config 4's stated size), file -> file through the warm pipe on %d context(s), once" % (BC, n_ctx)}
                    for trio in big_outs:
                        for pth in trio:
                            if pth and os.path.exists(pth):
                                os.unlink(pth)
                finally:
                    shutil.rmtree(big_work, ignore_errors=True)
            else:
                big = {"skipped": "not enough room for %d copies: %.0f GB free for the outputs in %s" % (BC, free_out / 1e9, work)}
        # ---- N ranks, N GPUs: the OTHER shape next to `value` (one input over N GPUs, two output files: flat by design) — K = N inputs,
        #      one per GPU, every rank its own pipe on its own device over its own files, all at once: the shape that scales with the
        #      GPUs (files fan out).  Same barriers, max over ranks.  (With per-rank visibility `value` already IS this shape.)
        if world > 1 and not per_rank and not args.device_only and args.inputs > 0:
            my_work = tempfile.mkdtemp(prefix="aqc_bench_k%d_" % rank, dir=base)
            try:
                my_paths = []
                for k, t in enumerate(texts):
                    pth = os.path.join(my_work, "R%d.rank%d.fq" % (k + 1, rank))
                    with open(pth, "wb") as f:
                        f.write(memoryview(t[0].array)[:t[1]])
                    my_paths.append(pth)
                my_outs = [(os.path.join(my_work, "R%d.rank%d.good.fq" % (k + 1, rank)), os.path.join(my_work, "R%d.rank%d.bad.fq" % (k + 1, rank)), None)
                           for k in range(len(texts))]
                my_pipe = capi.Pipe([eng], slots=args.slots)
                k_times = []
                for it in range(3):
                    eng.reset_stats()
                    for trio in my_outs:
                        for pth in trio:
                            if pth and os.path.exists(pth):
                                os.unlink(pth)
                    barrier()
                    t1 = time.perf_counter()
                    r_k = my_pipe.run(my_paths, my_outs, chunk_records=K, qc_sample=args.qc_sample)
                    if torch.cuda.is_available():
                        torch.cuda.synchronize()
                    if dist is not None:
                        dist.barrier()
                    dt = max_over_ranks(time.perf_counter() - t1)
                    assert not r_k.anomaly and int(r_k.records) == n_rec
                    if it:
                        k_times.append(dt)
                my_pipe.close()
                if rank == 0:
                    multi = {"inputs": world, "mreads_s": round(world * reads_per_gpu / min(k_times) / 1e6, 2), "seconds": round(min(k_times), 4), "runs": len(k_times),
                             "output_files": 2 * world * len(texts),
                             "what": "%d independent inputs of %.1f M reads each, one per rank / GPU, every rank its own pipe over its own files, all at once "
                                     "(barrier on both sides, slowest rank's time)" % (world, reads_per_gpu / 1e6)}
            finally:
                shutil.rmtree(my_work, ignore_errors=True)
        # ---- the same through gzip both ways: one-member inputs decoded by the host pool (the box's CPU quota is the bound),
        # .gz members built on the device
        gz_runs = args.gz_runs if args.gz_runs >= 0 else (3 if copies == 1 and shutil.which("gzip") else 0)
        if gz_runs > 0 and rank == 0 and not args.device_only:
            import subprocess
            gz_paths = [p + ".gz" for p in paths]
            jobs = [subprocess.Popen(["gzip", "-2", "-c", p], stdout=open(g, "wb")) for p, g in zip(paths, gz_paths)]
            for j in jobs:
                if j.wait() != 0:
                    raise RuntimeError("gzip failed")
            gouts = [(o[0] + ".gz", o[1] + ".gz", None) for o in outs]
            gz_before = (capi.C.c_uint64 * 4)()
            capi.load_library().aqc_gz_input_stats(capi.C.byref(gz_before))
            # (round 6: the device's share of the gunzip is the DEFAULT for inputs of this size — a decoder is started for every .gz
            #  input of >= 448 MiB, its buffers set up in the background, markers and CRC-32 resolved on the device; no environment
            #  override is set for the "default" runs below.  The host pool alone, AQC_GZ_DEVICE_IN=0, is timed next to it.)
            ts, ts_host = [], []
            for mode in ("host", "default"):
                os.environ.pop("AQC_GZ_DEVICE_MIN", None)
                os.environ.pop("AQC_GZ_DEVICE_IN", None)
                if mode == "host":
                    os.environ["AQC_GZ_DEVICE_IN"] = "0"
                else:
                    capi.load_library().aqc_gz_input_stats(capi.C.byref(gz_before))
                for it in range((gz_runs if mode == "default" else min(gz_runs, 2)) + 1):
                    reset_all()
                    for trio in gouts:
                        for pth in trio:
                            if pth and os.path.exists(pth):
                                os.unlink(pth)
                    t1 = time.perf_counter()
                    pr = pipe.run(gz_paths, gouts, gzip_in=[True] * len(gz_paths), gzip_out=True, gzip_level=2, chunk_records=K, qc_sample=args.qc_sample)
                    dt = time.perf_counter() - t1
                    assert not pr.anomaly and int(pr.records) == n_rec * copies
                    if mode == "default" and it == 0:
                        first_default = dt          # the pipe's FIRST device-assisted run: the decoder's buffers are set up during it
                    if it:
                        (ts if mode == "default" else ts_host).append(dt)
            os.environ.pop("AQC_GZ_DEVICE_IN", None)
            os.environ.pop("AQC_GZ_DEVICE_MIN", None)
            gzs = (capi.C.c_uint64 * 4)()
            capi.load_library().aqc_gz_input_stats(capi.C.byref(gzs))
            sec_all, sec_dev, by_all, by_dev = [int(a) - int(b) for a, b in zip(gzs, gz_before)]
            # plain text in -> .gz out (the output a third of the size: the two writers are no longer the bound)
            tz = []
            for it in range(gz_runs + 1):
                reset_all()
                for trio in gouts:
                    for pth in trio:
                        if pth and os.path.exists(pth):
                            os.unlink(pth)
                t1 = time.perf_counter()
                pz = pipe.run(paths, gouts, gzip_out=True, gzip_level=2, chunk_records=K, qc_sample=args.qc_sample)
                dtz = time.perf_counter() - t1
                assert not pz.anomaly and int(pz.records) == n_rec * copies
                if it:
                    tz.append(dtz)
            f2gz = {"mreads_s": round(pipe_reads / min(tz) / 1e6, 2), "seconds": round(min(tz), 4), "runs": len(tz), "thread_seconds_last_run": pz.breakdown()}
            # mreads_s is the DEFAULT path for an input of this size (round-4 advisory), which since round 6 is the pool AND the device;
            # the host pool alone stands beside it
            host_only = round(pipe_reads / min(ts_host) / 1e6, 2) if ts_host else None
            f2f_gz = {"mreads_s": round(pipe_reads / min(ts) / 1e6, 2), "seconds": round(min(ts), 4), "median_seconds": round(sorted(ts)[len(ts) // 2], 4), "runs": len(ts),
                      "path": "default for this input size, no environment override: one gzip member per file inflated by the host pool AND the GPU (groups of sections: "
                              "aqc_gunzip.cpp + aqc_gunzip_dev.hpp); the device resolves the markers and the CRC-32 of its sections itself and hands back text; "
                              ".gz members of the outputs built on the device",
                      "first_run_seconds": round(first_default, 4), "first_run_what": "the pipe's first .gz run with the device decoder (its buffers are set up in the background during it), untimed warm-up of the figure above",
                      "host_only_mreads_s": host_only, "host_only_seconds": round(min(ts_host), 4) if ts_host else None, "host_only_runs": len(ts_host),
                      "gunzip_sections": sec_all, "gunzip_sections_from_device": sec_dev, "gunzip_text_share_from_device": round(by_dev / max(1, by_all), 3),
                      "input_gz_gb": round(sum(os.path.getsize(g) for g in gz_paths) / 1e9, 3),
                      "output_gz_gb": round(sum(os.path.getsize(x) for trio in gouts for x in trio if x and os.path.exists(x)) / 1e9, 3),
                      "thread_seconds_last_run": pr.breakdown(), "cpu_quota": _cpu_quota()}
    finally:
        if work:
            shutil.rmtree(work, ignore_errors=True)
    if pipe is not None:
        pipe.close()

    if step_times:
        elapsed = sum(step_times)
        ms_per_step = 1000.0 * elapsed / len(step_times)
        value = pipe_reads / (elapsed / len(step_times)) / 1e6
    else:                       # --device-only: a profiling run, not the metric
        ms_per_step, value = dev_ms, dev_value

    wl = {"config3": "config3: ONE input of %d synthetic PE 2x150 bp pairs (%.1f M reads; %d x the per-GPU share of %d pairs) as two FASTQ files "
                     "(%.2f GB) -> four good / bad FASTQ files, seed 1003, overlap ~N(30,8), 3%% adapter read-through, defaults with -f 0 -t 0, "
                     "qc_sample %d" % (n_rec * copies, pipe_reads / 1e6, copies, n_rec, text_in * copies / 1e9, args.qc_sample),
          "config2": "config2: ONE input of %d synthetic SE 1x150 bp reads as one FASTQ file -> good / bad files, -f 5 -t 5, qc_sample %d" % (n_rec * copies, args.qc_sample),
          "config5": "config5 (plain text; gzip: tools/e2e_bench.py / --gz-runs): ONE input of %d synthetic PE 2x250 bp pairs + 17-base barcode/verify "
                     "prefix as two FASTQ files -> good / bad files, barcode mode, qc_sample %d" % (n_rec * copies, args.qc_sample)}[args.workload]
    traffic, traffic_src = hbm_traffic(args, n_rec, live=(rank == 0 and world == 1 and not args.device_only and not args.no_pmc))
    out = {
        "metric": ({"config3": "Mreads/s (paired 2x150 bp) end-to-end good/bad split", "config2": "Mreads/s (single-end 1x150 bp, BASELINE config 2) end-to-end good/bad split",
                    "config5": "Mreads/s (paired 2x250 bp + barcodes, BASELINE config 5 shape) end-to-end good/bad split"}[args.workload]
                   if step_times else "DEVICE STEP ONLY (--device-only profiling run, not the metric)"),
        "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": wl + "; step = file -> file: aqc_pipe_run (pread -> page-locked rings -> H2D -> framing -> filter / overlap / "
                                    "correction verdicts -> QC sampling -> good / bad text -> D2H -> one writer per output file)",
                   "pairs_per_gpu": args.pairs, "read_len": RL,
                   "parallelism": ("%d ranks, each its own pipe on its own GPU over its own %d-pair share of the input (the launcher shows a rank one device: one "
                                   "process cannot drive them all), no collective" % (world, n_rec)) if per_rank else
                                  ("one input, chunks of %d records dealt round robin over %d context(s) on device(s) %s by one process, no collective" % (K, n_ctx, dev_list)),
                   "text_in_gb": round(text_in * copies / 1e9, 3), "timing": "%d runs, each bracketed by barrier + device sync; outputs of the previous run unlinked in between (untimed)" % len(step_times)},
        "roofline": {"bound": "hbm", "kernel": "fast_filter_overlap_kernel (+ its deferral list kernel)", "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel_ms": round(k_ms, 4), "launches": int(klaunch[capi.K_FILTER_OVERLAP]),
                     "algorithmic_bytes_per_launch": n_rec * bytes_per_record,
                     "qc_stat_ms_per_call": round(float(kms[capi.K_QC_STAT]), 4),
                     "measured_in": "the device-step loop of this run (HIP events on the slot's stream)"},
        "device_step_mreads_s": round(dev_value, 2),
        "device_step_spans": {"ms_per_step": round(1000.0 * spans_elapsed / dsteps, 4), "mreads_s": round(reads_per_gpu * world / max(spans_elapsed / dsteps, 1e-9) / 1e6, 2),
                              "rebuilt_good_and_bad_text_gb_per_gpu": round(sum(spans_sizes) / 1e9, 4),
                              "what": "the device step as aqc_pipe_run issues it for plain-text outputs: aqc_reframe -> aqc_run -> aqc_qc_stat -> aqc_format_spans "
                                      "— good records that go out as their own bytes are not copied on the device (the file writers take them from the "
                                      "page-locked input buffer), only the bad / trimmed / corrected records are formatted"},
        "device_step_fused": fused,
        "device_step": {"ms_per_step": round(dev_ms, 4), "steps": dsteps, "text_in_gb_per_gpu": round(text_in / 1e9, 3),
                        "text_out_gb_per_gpu": round(text_out / 1e9, 3), "step_text_gb_s": round((text_in + text_out) / (dev_ms * 1e-3) / 1e9, 1),
                        "what": "text resident in HBM -> aqc_reframe -> aqc_run -> aqc_qc_stat -> aqc_format (no PCIe, no files), every rank on its own GPU: the step "
                                "aqc_pipe_run issues by default.  The cheaper steps (`device_step_spans`, `device_step_fused`) are opt-in: as defaults they lose "
                                "END TO END on a host bound by its two file writers (profiles/r05_spans_ab.txt, r06_spans_assemble_ab.txt)"},
        "pinned_to_pinned_mreads_s": pinned["mreads_s"] if pinned else None,
        "pinned_to_pinned": pinned, "file_to_file": f2f, "file_to_file_100M": big, "file_to_file_gz": f2f_gz, "file_to_gz": f2gz,
        "multi_input_file_to_file_mreads_s": multi["mreads_s"] if multi else None, "multi_input_file_to_file": multi,
        "good_reads_frac": round(good_frac, 5),
        "gen_s": round(t_gen, 1), "text_render_s": round(t_txt, 1), "first_upload_s": round(t_up, 3),
        # `value` is the contract's figure: reads over the MEAN of the timed steps.  The boxes differ by more than a round changes
        # (the same code: 48 - 56 Mreads/s over four boxes in round 5), and single steps by +-8 %: the median and the best step of
        # THIS run stand beside it, with the box's name
        "value_median": round(pipe_reads / sorted(step_times)[len(step_times) // 2] / 1e6, 3) if step_times else None,
        "value_best": round(pipe_reads / min(step_times) / 1e6, 3) if step_times else None,
        "host": dict({"cpus": os.cpu_count(), "cpu_quota": _cpu_quota()}, **_host_identity()),
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
        same = bool(np.array_equal(oe.fetch_results(0), res_gpu[:m]))
        out["cpu_baseline"] = {"value": round(2 * m / tc / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
                               "sample": "first %d pairs of the same input, oracle/aqc_oracle.c (scalar C restatement of "
                                         "preprocesser.py:411-631), filter+overlap+correction only (no text framing / formatting), %.1f s" % (m, tc),
                               "matches_gpu": same, "host_cpus": os.cpu_count()}
        # the same port on every host core (independent slices, one oracle context per thread; the C call releases the
        # GIL): SURVEY.md §8d asks for the 1-core and the all-cores figure side by side
        try:
            from concurrent.futures import ThreadPoolExecutor
            T = max(1, min(os.cpu_count() or 1, 64))
            per = max(1000, min(100_000, args.pairs // T))
            engines_o = []
            for k in range(T):
                lo, hi = k * per, (k + 1) * per
                e = oracle.OracleEngine()
                e.set_config(cfg)
                e.upload(0, capi.Batch.from_matrices(d["seq1"][lo:hi], d["qual1"][lo:hi], d["len1"][lo:hi],
                                                     d["seq2"][lo:hi], d["qual2"][lo:hi], d["len2"][lo:hi]))
                engines_o.append(e)
            ta = time.perf_counter()
            with ThreadPoolExecutor(max_workers=T) as ex:
                list(ex.map(lambda e: e.run(0), engines_o))
            ta = time.perf_counter() - ta
            out["cpu_baseline"]["all_cores"] = {"value": round(2 * per * T / ta / 1e6, 3), "unit": "Mreads/s", "cores": T,
                                                "sample": "%d threads x %d pairs, %.2f s (cgroup CPU quota: %s)" % (T, per, ta, _cpu_quota())}
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
            gflags = res_gpu[:mp]["flag"]
            out["cpu_baseline"]["cpython_standin"] = {
                "value": round(2 * mp / tp / 1e6, 5), "unit": "Mreads/s", "cores": 1,
                "sample": "first %d pairs, oracle/pyloop.py (pure-Python restatement of the reference loop), %.1f s" % (mp, tp),
                "matches_gpu": bool(all(int(p["flag"]) == int(f) for p, f in zip(py, gflags)))}
        except Exception as e:
            out["cpu_baseline"]["cpython_standin"] = {"error": str(e)}
    if rank == 0:
        print(json.dumps(out))
    for t in texts:
        t[0].free()
    for e in engines[1:]:
        e.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _host_identity():
    """which box this ran on (numbers from different boxes differ by up to 25 % on the bandwidth-bound kernels): host name, and the
    GPU's unique id where rocm-smi is at hand"""
    import socket
    import subprocess
    out = {"hostname": socket.gethostname()}
    try:
        txt = subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout
        ids = [ln.split(":")[-1].strip() for ln in txt.splitlines() if "Unique ID" in ln]
        if ids:
            out["gpu_unique_id"] = ids[0] if len(ids) == 1 else ids
    except Exception:
        pass
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    out["cpu_model"] = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return out


def _cpu_quota():
    """CPUs' worth of run time the cgroup allows this container (cpu.max), or None when unlimited / unknown: the MI355X boxes
    show 256 hardware threads but run under a quota, which is what bounds every host-side stage (gzip above all)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        return None if q == "max" else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
