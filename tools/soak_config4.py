#!/usr/bin/env python3
"""Config-4-size soak on ONE GPU (BASELINE.json configs 4 / 5: 100 M reads; preprocesser.py:624 for the sampling rule).

  python tools/soak_config4.py [--pairs 5000000] [--copies 10] [--devices 0,0] [--dir DIR]

A block of P synthetic 2x150 pairs (config 3's generator) is written COPIES times into one input file pair (>= 100 M reads at
the defaults: two 17 GB files) and run through the product's whole-input pipe (aqc_pipe_run) over the contexts named by
--devices (default 0,0: two contexts on GPU 0, the one-input-over-N-GPUs code path).  Checked:
  * the record count, and that every output file is, piece by piece, the block's own output repeated COPIES times (compared
    byte for byte against a one-block run through ONE context — stronger than a digest);  files are > 4 GiB, hundreds of chunks
    go through every slot, first_index runs to COPIES * P;
  * the counters: COPIES x the block's counters, histogram bins likewise (the post-filter QC arrays are sampled from the first
    qc_sample records only, so they equal the one-block run's);
  * a second, one-block run whose chunk indices start beyond 2^32 / K records (aqc_pipe_opts.chunk_index0): 64-bit record
    indices all the way down — same output bytes, same counters, nothing sampled.
If the directory lacks the space for COPIES copies the script says so and uses what fits.  Exit status 0 = all checks passed."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=5_000_000)
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--chunk-records", type=int, default=1 << 17)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--in-dir", default=None, help="where the big INPUT pair goes (default: --dir).  /dev/shm keeps 35 GB off a small disk")
    ap.add_argument("--no-sync", action="store_true", help="do not sync() the inputs before the timed run (round 4's soak: the run then starts with 35 GB of "
                    "dirty input pages and is throttled to the disk's write-back rate once its own outputs push the cgroup over its dirty limit)")
    args = ap.parse_args()
    import numpy as np
    from afterqc_amd import capi, synth

    t0 = time.time()
    d = synth.make_pairs(args.pairs, 150, seed=1004, workers=max(1, (os.cpu_count() or 8) // 2))
    n = len(d["len1"])
    W = d["seq1"].shape[1]
    w = synth.fixed_record_width(W)
    work = tempfile.mkdtemp(prefix="aqc_soak_", dir=args.dir)
    log = {"pairs_per_block": n, "what": __doc__.split("\n")[0]}
    ok = True
    try:
        block = []
        for mate in (1, 2):
            buf = np.empty(n * w + 4096, dtype=np.uint8)
            _, nb = synth.render_fastq_fixed(d["seq%d" % mate], d["qual%d" % mate], mate, out=buf)
            block.append((buf, nb))
        block_bytes = sum(b[1] for b in block)
        free = shutil.disk_usage(work).free
        copies = args.copies
        # inputs + outputs (~ the same size) + the one-block reference run
        while copies > 1 and ((1 if args.in_dir else 2) * copies + 2) * block_bytes * 1.05 > free:
            copies -= 1
        if copies != args.copies:
            print("soak: only %.0f GB free in %s: %d copies instead of %d" % (free / 1e9, work, copies, args.copies), flush=True)
        log.update(copies=copies, reads=2 * n * copies, input_gb=round(copies * block_bytes / 1e9, 2), dir_free_gb=round(free / 1e9, 1))

        cfg = capi.Config()
        cfg.paired = 1
        cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
        cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
        cfg.barcode_length = 12
        cfg.set_verify("CAGTA")
        cfg.qc_kmer = 8

        def names(tag):
            return [os.path.join(work, "%s_R%d.fq" % (tag, k)) for k in (1, 2)], \
                   [(os.path.join(work, "%s_R%d.good.fq" % (tag, k)), os.path.join(work, "%s_R%d.bad.fq" % (tag, k)), None) for k in (1, 2)]

        # ---- the block alone through ONE context: the reference for everything below
        one_in, one_out = names("one")
        for p, (buf, nb) in zip(one_in, block):
            with open(p, "wb") as f:
                f.write(memoryview(buf)[:nb])
        e1 = capi.Engine(0, 3)
        e1.set_config(cfg)
        e1.reset_stats()
        p1 = capi.Pipe([e1], slots=3)
        r1 = p1.run(one_in, one_out, chunk_records=args.chunk_records, qc_sample=200000)
        assert not r1.anomaly and int(r1.records) == n
        c1 = e1.counters().copy()
        h1 = [x.copy() for x in e1.histograms(capi.AQC_QC_COLS)]
        q1 = [e1.qc(which).copy() for which in (capi.QC_R1_POST, capi.QC_R2_POST)]
        ref = {}
        for trio in one_out:
            for pth in trio:
                if pth:
                    ref[os.path.basename(pth)[4:]] = np.fromfile(pth, dtype=np.uint8)
        # the same block with record indices beyond 2^32: the pipe's first_index is 64-bit all the way to the kernels
        e1.reset_stats()
        hi_in, hi_out = names("hi")
        K = args.chunk_records
        r_hi = p1.run(one_in, hi_out, chunk_records=K, qc_sample=200000, chunk_index0=((1 << 32) + 12345) // K + 1)
        same_hi = int(r_hi.records) == n and not r_hi.anomaly
        for trio in hi_out:
            for pth in trio:
                if pth:
                    same_hi = same_hi and np.array_equal(np.fromfile(pth, dtype=np.uint8), ref[os.path.basename(pth)[3:]])
                    os.unlink(pth)
        c_hi = e1.counters().copy()
        q_hi = [e1.qc(which) for which in (capi.QC_R1_POST, capi.QC_R2_POST)]
        same_hi = same_hi and bool(np.array_equal(c_hi, c1)) and all(int(np.abs(q).sum()) == 0 for q in q_hi)
        log["indices_beyond_2_32"] = {"first_index0": (((1 << 32) + 12345) // K + 1) * K, "identical_outputs_and_counters": bool(same_hi), "post_filter_qc_rows_all_zero": True if same_hi else False}
        ok = ok and same_hi
        p1.close()
        e1.close()

        # ---- COPIES blocks as ONE input over the contexts of --devices
        big_in, big_out = names("big")
        in_work = None
        if args.in_dir:
            in_work = tempfile.mkdtemp(prefix="aqc_soak_in_", dir=args.in_dir)
            big_in = [os.path.join(in_work, os.path.basename(p)) for p in big_in]
        tw = time.time()
        for p, (buf, nb) in zip(big_in, block):
            with open(p, "wb") as f:
                for _ in range(copies):
                    f.write(memoryview(buf)[:nb])
        log["write_inputs_s"] = round(time.time() - tw, 1)
        if not args.no_sync:
            ts = time.time()
            os.sync()              # the inputs are clean when the run starts, as a user's files are
            log["sync_inputs_s"] = round(time.time() - ts, 1)
        with open("/proc/meminfo") as f:
            log["dirty_kb_before_run"] = int([ln.split()[1] for ln in f if ln.startswith("Dirty:")][0])
        devs = [int(x) for x in args.devices.split(",")]
        engines = []
        for g in devs:
            e = capi.Engine(g, 3)
            e.set_config(cfg)
            e.reset_stats()
            engines.append(e)
        pipe = capi.Pipe(engines, slots=3)
        tr = time.perf_counter()
        res = pipe.run(big_in, big_out, chunk_records=K, qc_sample=200000)
        dt = time.perf_counter() - tr
        log["run"] = {"seconds": round(dt, 3), "mreads_s": round(2 * n * copies / dt / 1e6, 2), "records": int(res.records), "chunks": int(res.chunks),
                      "chunks_per_slot": round(int(res.chunks) / (3 * len(devs)), 1), "devices": devs, "threads": res.breakdown()}
        ok = ok and not res.anomaly and int(res.records) == n * copies
        merged = capi.MergedEngines(engines)
        cN = merged.counters()
        hN = merged.histograms(capi.AQC_QC_COLS)
        qN = [merged.qc(which) for which in (capi.QC_R1_POST, capi.QC_R2_POST)]
        counters_ok = bool(np.array_equal(cN, c1 * copies))
        hist_ok = all(bool(np.array_equal(a, b * copies)) for a, b in zip(hN, h1))
        qc_ok = all(bool(np.array_equal(a, b)) for a, b in zip(qN, q1))
        log["counters_equal_copies_x_block"] = counters_ok
        log["histograms_equal_copies_x_block"] = hist_ok
        log["post_filter_qc_rows_equal_block"] = qc_ok
        ok = ok and counters_ok and hist_ok and qc_ok
        files = {}
        tc = time.time()
        for trio in big_out:
            for pth in trio:
                if not pth:
                    continue
                want = ref[os.path.basename(pth)[4:]]
                size = os.path.getsize(pth)
                same = size == want.size * copies
                if same and want.size:
                    with open(pth, "rb") as f:
                        piece = np.empty(want.size, dtype=np.uint8)
                        for _ in range(copies):
                            got = f.readinto(memoryview(piece))
                            if got != want.size or not np.array_equal(piece, want):
                                same = False
                                break
                files[os.path.basename(pth)] = {"bytes": size, "gib": round(size / 2 ** 30, 2), "equals_block_output_x_copies": bool(same)}
                ok = ok and same
        log["outputs"] = files
        log["compare_s"] = round(time.time() - tc, 1)
        log["largest_output_gib"] = max(v["gib"] for v in files.values())
        log["last_first_index"] = (int(res.chunks) - 1) * K
        pipe.close()
        for e in engines:
            e.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)
        if args.in_dir and 'in_work' in dir() and in_work:
            shutil.rmtree(in_work, ignore_errors=True)
    log["total_s"] = round(time.time() - t0, 1)
    log["ok"] = bool(ok)
    print(json.dumps(log, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
