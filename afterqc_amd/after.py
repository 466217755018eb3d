#!/usr/bin/env python3
"""`after.py`-compatible command line for the MI355X engine.

Same 38 options, defaults and post-processing as the reference's CLI (after.py:14-93, 196-221) and
the same output layout (preprocesser.py:285-371), so `python -m afterqc_amd.after -1 R1.fq -2 R2.fq`
drops in for `python after.py -1 R1.fq -2 R2.fq`.  Differences by design: Python 3; one GPU context
per file pair instead of one OS process per file (after.py:168-171) — files of a directory are
spread round-robin over the visible GPUs; the debubble detector pre-pass (debubble.py, PIL) is out
of scope: `--debubble` consumes an existing <debubble_dir>/circles.csv exactly like `-1/-2` mode
does upstream (after.py:207-212, preprocesser.py:235-236).
"""
import copy
import os
import sys
import time
from optparse import OptionParser

AFTERQC_VERSION = "0.9.6"
USAGE = ("Automatic Filtering, Trimming, Error Removing and Quality Control for Illumina fastq data \n\n"
         "Simplest usage:\ncd to the folder containing your fastq data, run <python after.py>")

# (short, long, kwargs) — one row per option of after.py:17-92, same dest / default / type
OPTIONS = [
    ("-1", "--read1_file", dict(help="file name of read1, required. If input_dir is specified, then this arg is ignored.")),
    ("-2", "--read2_file", dict(default=None, help="file name of read2, if paired. If input_dir is specified, then this arg is ignored.")),
    ("-7", "--index1_file", dict(default=None, help="file name of 7' index. If input_dir is specified, then this arg is ignored.")),
    ("-5", "--index2_file", dict(default=None, help="file name of 5' index. If input_dir is specified, then this arg is ignored.")),
    ("-d", "--input_dir", dict(default=None, help="the input dir to process automatically. If read1_file are input_dir are not specified, then current dir (.) is specified to input_dir")),
    ("-g", "--good_output_folder", dict(default="good", help="the folder to store good reads, by default it is named 'good', in the current directory")),
    ("-b", "--bad_output_folder", dict(default=None, help="the folder to store bad reads, by default it is named 'bad', in the same folder as good_output_folder")),
    ("-r", "--report_output_folder", dict(default=None, help="the folder to store QC reports, by default it is named 'QC', in the same folder as good_output_folder")),
    ("", "--read1_flag", dict(default="R1", help="specify the name flag of read1, default is R1, which means a file with name *R1* is read1 file")),
    ("", "--read2_flag", dict(default="R2", help="specify the name flag of read2, default is R2, which means a file with name *R2* is read2 file")),
    ("", "--index1_flag", dict(default="I1", help="specify the name flag of index1, default is I1, which means a file with name *I1* is index2 file")),
    ("", "--index2_flag", dict(default="I2", help="specify the name flag of index2, default is I2, which means a file with name *I2* is index2 file")),
    ("-f", "--trim_front", dict(default=-1, type="int", help="number of bases to be trimmed in the head of read. -1 means auto detect")),
    ("-t", "--trim_tail", dict(default=-1, type="int", help="number of bases to be trimmed in the tail of read. -1 means auto detect")),
    ("", "--trim_pair_same", dict(default="true", help="use same trimming configuration for read1 and read2 to keep their sequence length identical, default is true")),
    ("-q", "--qualified_quality_phred", dict(default=15, type="int", help="the quality value that a base is qualifyed. Default 15 means phred base quality >=Q15 is qualified.")),
    ("-u", "--unqualified_base_limit", dict(default=60, type="int", help="if exists more than unqualified_base_limit bases that quality is lower than qualified quality, then this read/pair is bad. Default is 60")),
    ("-p", "--poly_size_limit", dict(default=35, type="int", help="if exists one polyX(polyG means GGGGGGGGG...), and its length is >= poly_size_limit, then this read/pair is bad. Default is 35")),
    ("-a", "--allow_mismatch_in_poly", dict(default=2, type="int", help="the count of allowed mismatches when detection polyX. Default 2 means allow 2 mismatches for polyX detection")),
    ("-n", "--n_base_limit", dict(default=5, type="int", help="if exists more than maxn bases have N, then this read/pair is bad. Default is 5")),
    ("-s", "--seq_len_req", dict(default=35, type="int", help="if the trimmed read is shorter than seq_len_req, then this read/pair is bad. Default is 35")),
    ("", "--debubble", dict(action="store_true", default=False, help="specify whether apply debubble algorithm to remove the reads in the bubbles. Default is False")),
    ("", "--debubble_dir", dict(default="debubble", help="specify the folder to store output of debubble algorithm, default is debubble")),
    ("", "--draw", dict(default="on", help="specify whether draw the pictures or not, when use debubble or QC. Default is on")),
    ("", "--barcode", dict(default="on", help="specify whether deal with barcode sequencing files, default is on, which means all files with barcode_flag in filename will be treated as barcode sequencing files")),
    ("", "--barcode_length", dict(default=12, type="int", help="specify the designed length of barcode")),
    ("", "--barcode_flag", dict(default="barcode", help="specify the name flag of a barcoded file, default is barcode, which means a file with name *barcode* is a barcoded file")),
    ("", "--barcode_verify", dict(default="CAGTA", help="specify the verify sequence of a barcode which is adjunct to the barcode")),
    ("", "--store_overlap", dict(default="off", help="specify whether store only overlapped bases of the good reads")),
    ("", "--overlap_output_folder", dict(default=None, help="the folder to store only overlapped bases of the good reads")),
    ("", "--qc_only", dict(action="store_true", default=False, help="if qconly is true, only QC result will be output, this can be much fast")),
    ("", "--qc_sample", dict(default=200000, type="int", help="sample up to qc_sample reads when do QC, 0 means sample all reads. Default is 200,000")),
    ("", "--qc_kmer", dict(default=8, type="int", help="specify the kmer length for KMER statistics for QC, default is 8")),
    ("", "--no_correction", dict(action="store_true", default=False, help="disable base correction for mismatched base pairs in overlapped areas")),
    ("", "--mask_mismatch", dict(action="store_true", default=False, help="set the qual num to 0 for mismatched base pairs in overlapped areas to mask them out")),
    ("", "--no_overlap", dict(action="store_true", default=False, help="disable overlap analysis (usually much faster with this option)")),
    ("-z", "--gzip", dict(action="store_true", default=False, help="force gzip compression for output, even the input is not gzip compressed")),
    ("", "--compression", dict(type="int", default=2, help="set compression level (0~9) for gzip output, default is 2 (0 = best speed, 9 = best compression).  Here: 1-9 all use the GPU's encoder (ratio about zlib level 2-3); 0 writes stored members; AQC_GZ_DEVICE=0 builds the files on the host at a speed / ratio that follows the level")),
]


def parseBool(s):
    """util.parseBool (util.py:29-34)"""
    return s.lower() in ("true", "yes", "on")


def parseCommand(argv=None):
    parser = OptionParser(usage=USAGE, version=AFTERQC_VERSION)
    for short, long_, kw in OPTIONS:
        names = [n for n in (short, long_) if n]
        parser.add_option(*names, dest=long_[2:], **kw)
    return parser.parse_args(argv)


def finalize_options(options):
    """The option post-processing of after.main (after.py:195-201)."""
    options.version = AFTERQC_VERSION
    options.trim_pair_same = parseBool(options.trim_pair_same)
    options.draw = parseBool(options.draw)
    options.store_overlap = parseBool(options.store_overlap)
    options.trim_front2 = options.trim_front
    options.trim_tail2 = options.trim_tail
    return options


def matchFlag(filename, flag):
    """after.py:95-99"""
    if flag[-1:] in (".", "_", "-"):
        return flag in filename
    return any((flag + sep) in filename for sep in (".", "_", "-"))


def collect_dir_jobs(folder, options):
    """The file pairing of processDir (after.py:101-166): one options object per R1 file."""
    from . import fastq
    jobs = []
    if not os.path.isdir(folder):
        return jobs
    for f in os.listdir(folder):
        path = os.path.join(folder, f)
        if os.path.isdir(path) or not fastq.isFastq(f) or f.startswith("Undetermined"):
            continue
        if not matchFlag(f, options.read1_flag):
            continue
        print(f)
        opt = copy.copy(options)
        opt.read1_file = path
        for attr, flag in (("read2_file", options.read2_flag), ("index1_file", options.index1_flag),
                           ("index2_file", options.index2_flag)):
            mate = path.replace(options.read1_flag, flag)
            if os.path.exists(mate):
                setattr(opt, attr, mate)
        if options.barcode_flag in f and parseBool(options.barcode):
            opt.barcode = True
            opt.trim_front = 0
            opt.trim_front2 = 0
        else:
            opt.barcode = False
        jobs.append(opt)
    return jobs


def processOptions(options, engine=None, device=0, devices=None):
    """after.processOptions (after.py:173-175).  `devices`: the GPUs ONE input is dealt over (chunks round robin, host-side
    merge of the statistics); a single file pair given with -1 / -2 uses every visible GPU."""
    from . import preprocesser
    flt = preprocesser.seqFilter(options, engine=engine, device=device, devices=devices)
    return flt.run()


def visible_devices():
    """GPUs this run may use: AQC_DEVICES=0,2,3 narrows the list (default: all that aqc_device_count() reports)"""
    from . import capi
    n = max(1, capi.load_library().aqc_device_count())
    env = os.environ.get("AQC_DEVICES")
    if env:
        devs = [int(x) for x in env.split(",") if x.strip() != ""]
        bad = [d for d in devs if d < 0 or d >= n]
        if bad or not devs:
            raise SystemExit("AQC_DEVICES=%s: device index out of range (this machine shows %d GPU%s)" % (env, n, "" if n == 1 else "s"))
        # (an index may be listed more than once: that many contexts on the one device — how a single-GPU box exercises the
        #  one-input-over-N-contexts path)
        return devs
    return list(range(n))


def processDir(folder, options, engine_factory=None, n_workers=None):
    """after.processDir (after.py:101-171): upstream forks one process per file pair; here one worker thread per
    visible GPU (one context each) takes the file pairs off a shared queue, so N GPUs filter N pairs at a time.
    `engine_factory(device)` lets the tests inject an engine."""
    import queue
    import threading
    jobs = collect_dir_jobs(folder, options)
    if not jobs:
        print("no read files to run with, do you call the program correctly?")
        print("see -h for help")
        return []
    # one worker per listed device (AQC_DEVICES may list a device more than once: that many workers share it)
    devs = [0] if engine_factory else visible_devices()
    if n_workers is None:
        n_workers = len(devs)
    n_workers = max(1, min(n_workers, len(jobs)))
    todo = queue.Queue()
    for k, opt in enumerate(jobs):
        todo.put((k, opt))
    stats = [None] * len(jobs)
    errors = []

    def worker(device):
        while True:
            try:
                k, opt = todo.get_nowait()
            except queue.Empty:
                return
            try:
                stats[k] = processOptions(opt, engine=engine_factory(device) if engine_factory else None, device=device)
            except BaseException as e:          # report after every worker has finished its files
                errors.append((opt.read1_file, e))

    threads = [threading.Thread(target=worker, args=(devs[d % len(devs)],)) for d in range(n_workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("; ".join("%s: %s" % (f, e) for f, e in errors))
    return stats


def main(argv=None):
    t0 = time.time()
    (options, args) = parseCommand(argv)
    finalize_options(options)
    if options.input_dir is None and options.read1_file is None:
        print('specify current dir as input dir')
        options.input_dir = "."
    if options.input_dir is not None:
        processDir(options.input_dir, options)
    else:
        if options.barcode_flag in options.read1_file and parseBool(options.barcode):
            options.barcode = True
            options.trim_front = 0      # barcoded reads are not trimmed at the front (after.py:217-219)
            options.trim_front2 = 0
        else:
            options.barcode = False
        processOptions(options, devices=visible_devices())
    print('Time used: ' + str(time.time() - t0))


if __name__ == "__main__":
    main()
