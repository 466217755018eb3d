#!/usr/bin/env python3
"""File-to-file throughput of the drop-in CLI path: what `python -m afterqc_amd.after -1 R1.fq -2 R2.fq` does, timed.

  python tools/e2e_bench.py --pairs 5000000                  # whole-input pipe (the default path)
  python tools/e2e_bench.py --pairs 2000000 --mode text      # serial chunk loop
  python tools/e2e_bench.py --pairs 200000 --mode host       # host framing / Python writer (cross-check path)
  python tools/e2e_bench.py --pairs 2000000 --config5 --gz   # 2x250 + barcodes + bubbles, .gz in -> .gz out

Writes config-3 style R1/R2 FASTQ files to --dir, runs afterqc_amd.preprocesser.seqFilter on them exactly as
`python -m afterqc_amd.after -1 R1.fq -2 R2.fq -f 0 -t 0` would, and prints one JSON line with the wall time of the
pre-filter sampling pass, of pass 2 (read -> filter -> write) and of the whole run."""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2_000_000)
    ap.add_argument("--mode", default="pipe", choices=["pipe", "text", "host"],
                    help="pipe: whole-input pipe (C++ threads, the default path); text: serial chunk loop; host: host framing / Python writer")
    ap.add_argument("--chunk-mb", type=int, default=64)
    ap.add_argument("--chunk-records", type=int, default=1 << 17)
    ap.add_argument("--devices", default="0", help="GPUs the input is dealt over, e.g. 0,1,2,3")
    ap.add_argument("--dir", default="/tmp/aqc_e2e")
    ap.add_argument("--single", action="store_true")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--reuse", action="store_true", help="use the input files a previous --keep run left in --dir (same --pairs / flavour) instead of making them again")
    ap.add_argument("--gz", action="store_true", help="gzip input (and therefore, like upstream, gzip output): ONE gzip member per file, "
                    "made with `gzip -<level>` — what real-world .fq.gz files are")
    ap.add_argument("--gz-level", type=int, default=6, help="level of the single-member .gz inputs (gzip's default: 6)")
    ap.add_argument("--bgzf", action="store_true", help="with --gz: inputs as BGZF-style independent 64 KiB members instead (the pipe's own "
                    "writer's format; member-parallel inflate)")
    ap.add_argument("--bz2", action="store_true", help="bzip2 inputs (fastq.py:25-26): plain text first, then the bzip2 program; outputs are plain text")
    ap.add_argument("--config5", action="store_true", help="config-5 flavour: 2x250 bp + 17 bp barcode/verify prefix, file names with 'barcode', --debubble with a circles.csv")
    args = ap.parse_args()
    from afterqc_amd import after, preprocesser, synth
    os.makedirs(args.dir, exist_ok=True)
    ext = ".fq.gz" if args.gz else ".fq.bz2" if args.bz2 else ".fq"
    stem = "barcode_" if args.config5 else ""
    r1, r2 = os.path.join(args.dir, stem + "R1" + ext), os.path.join(args.dir, stem + "R2" + ext)
    t = time.perf_counter()
    have = args.reuse and os.path.exists(r1) and (args.single or os.path.exists(r2))
    d = None if have else synth.make_pairs(args.pairs, 250 if args.config5 else 150, seed=1005 if args.config5 else 1003,
                         workers=max(1, (os.cpu_count() or 8) // 2))
    if args.config5 and not have:
        d = synth.add_barcodes(d, 1005 + 7)
        os.makedirs(os.path.join(args.dir, "D"), exist_ok=True)
        with open(os.path.join(args.dir, "D", "circles.csv"), "w") as f:
            # names carry tile 1101 -> int(tile[1:]) = 101, lane 1, x = record index, y = index * 7919 % 100000
            f.write("x,y,radius,lane,tile\n")
            for k in range(8):
                f.write("%r,%r,%r,1,101\n" % (250000.0 * (k + 1), 50000.0, 3000.0 + 100.0 * k))
    if have:
        pass
    elif args.gz and not args.bgzf:
        # single-member .gz: plain text first, then the gzip program (both mates at once)
        import subprocess
        jobs = []
        for path, mate in ((r1, 1),) + (() if args.single else ((r2, 2),)):
            plain = path[:-3]
            synth.write_fastq_fixed(plain, d["seq%d" % mate], d["qual%d" % mate], mate)
            jobs.append((subprocess.Popen(["gzip", "-%d" % args.gz_level, "-c", plain], stdout=open(path, "wb")), plain))
        for pr, plain in jobs:
            assert pr.wait() == 0
            os.unlink(plain)
    elif args.bz2:
        import subprocess
        jobs = []
        for path, mate in ((r1, 1),) + (() if args.single else ((r2, 2),)):
            plain = path[:-4]
            synth.write_fastq_fixed(plain, d["seq%d" % mate], d["qual%d" % mate], mate)
            jobs.append((subprocess.Popen(["bzip2", "-k", "-f", plain]), plain))
        for pr, plain in jobs:
            assert pr.wait() == 0
            os.unlink(plain)
    else:
        synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
        if not args.single:
            synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    del d
    os.sync()          # the inputs are clean when the run starts, as a user's files are (dirty input pages count against the run's own write-back budget)
    gen_s = time.perf_counter() - t
    argv = ["-1", r1] + ([] if args.single else ["-2", r2]) + ["-f", "0", "-t", "0", "-g", os.path.join(args.dir, "good"),
                                                              "-b", os.path.join(args.dir, "bad"), "-r", os.path.join(args.dir, "QC")]
    if args.config5:
        argv += ["--debubble", "--debubble_dir", os.path.join(args.dir, "D")]
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    options.barcode = bool(args.config5)       # what after.py:215-221 decides from the file name
    if args.config5:
        options.trim_front = options.trim_front2 = 0
    flt = preprocesser.seqFilter(options, use_text_path=args.mode != "host", use_pipe=args.mode == "pipe", chunk_bytes=args.chunk_mb << 20,
                                 chunk_records=args.chunk_records, devices=[int(x) for x in args.devices.split(",")])
    t = time.perf_counter()
    stat = flt.run()
    wall = time.perf_counter() - t
    reads = args.pairs * (1 if args.single else 2)
    in_bytes = os.path.getsize(r1) + (0 if args.single else os.path.getsize(r2))
    s = stat["afterqc_main_summary"]
    out = {"mode": args.mode, "gz": ("bgzf" if args.bgzf else "single member, gzip -%d" % args.gz_level) if args.gz else False, "pairs": args.pairs, "reads": reads, "input_bytes": in_bytes, "gen_s": round(gen_s, 1),
           "wall_s": round(wall, 3), "pass1_s": round(flt.timing["pass1_s"], 3), "pass2_s": round(flt.timing["pass2_s"], 3),
           "e2e_mreads_s": round(reads / wall / 1e6, 3), "pass2_mreads_s": round(reads / flt.timing["pass2_s"] / 1e6, 3),
           "pass2_input_gb_s": round(in_bytes / flt.timing["pass2_s"] / 1e9, 3), "pass2_cores_busy": round(flt.timing["pass2_cpu_s"] / flt.timing["pass2_s"], 1),
           "init_s": round(flt.timing.get("init_s", 0), 3), "stats_s": round(flt.timing.get("stats_s", 0), 3), "report_s": round(flt.timing.get("report_s", 0), 3),
           "close_s": round(flt.timing.get("close_s", 0), 3),
           "good_reads": s["good_reads"], "bad_reads": s["bad_reads"], "text_path": flt.text_path, "used_pipe": flt.used_pipe,
           "config5": args.config5, "devices": args.devices, "pipe_threads": flt.timing.get("pipe_threads")}
    print(json.dumps(out))
    if not args.keep:
        shutil.rmtree(args.dir, ignore_errors=True)
    else:
        for sub in ("good", "bad", "QC"):          # the inputs stay (--reuse), the outputs go: the next run must not start on gigabytes of dirty pages
            shutil.rmtree(os.path.join(args.dir, sub), ignore_errors=True)


if __name__ == "__main__":
    main()
