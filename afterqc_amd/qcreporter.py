"""QC/<read1 basename>.html — the report the reference writes next to the stats JSON (qcreporter.py:36-55 page skeleton,
qualitycontrol.py:158-322 figures, preprocesser.py:677-700,771-772,785-830 figure list).

Same page: a menu, the summary table, then one Plotly figure per entry with the reference's titles and div ids
(`filter_stat`, `error_matrix`, `overlap_stat`, `r{1,2}_{pre,post}_{quality,content,gc,discontinuity,sb}`), fed from the
integers the device accumulated (QualityControl objects) and the stats dictionary.  Built here as data (one dict per
figure: traces + layout, serialised with json) rather than by string concatenation; numbers are the same float64 values the
stats JSON holds.  The page loads plotly.js from the CDN exactly like upstream's.
"""
import json

from .qc import ALL_BASES

_BASE_COLOURS = {"A": "rgba(255,0,0,0.5)", "T": "rgba(128,0,128,0.5)", "C": "rgba(0,255,0,0.5)", "G": "rgba(0,0,255,0.5)"}
_COMP = {"A": "T", "T": "A", "C": "G", "G": "C", "a": "t", "t": "a", "c": "g", "g": "c", "N": "N"}


def div_id(title):
    """anchor name of a figure title (formatDivID, qcreporter.py:3-7)"""
    for ch in " ./":
        title = title.replace(ch, "-")
    return title


def human(num):
    """1234567 -> '1.177 M' (formatNumber, qcreporter.py:9-20: powers of 1024, three decimals)"""
    num = float(num)
    order = 0
    while num > 1024.0:
        order += 1
        num /= 1024.0
    return str(int(num)) if order == 0 else "%0.3f %s" % (num, ["", "K", "M", "G", "T", "P"][order])


def _lines(xs, series, colours, names):
    return [{"x": xs, "y": list(y), "name": n, "mode": "lines", "line": {"color": c, "width": 1}} for y, c, n in zip(series, colours, names)]


class Figure:
    def __init__(self, title, div, traces, layout):
        self.title, self.div, self.traces, self.layout = title, div, traces, layout

    def script(self):
        if self.traces is None:
            return ""
        return "var data=%s;\nvar layout=%s;\nPlotly.newPlot('%s', data, layout);\n" % (json.dumps(self.traces), json.dumps(self.layout), self.div)


# ---- the five per-QualityControl figures (qualitycontrol.py:158-270) ---------------------------------------------------------
def quality_figure(qc, div, title):
    n = qc.readLen
    xs = list(range(n))
    series = [qc.baseMeanQual[b][:n].tolist() for b in ALL_BASES] + [qc.meanQual[:n].tolist()]
    traces = _lines(xs, series, [_BASE_COLOURS[b] for b in ALL_BASES] + ["rgba(20,20,20,255)"], list(ALL_BASES) + ["mean"])
    return Figure(title, div, traces, {"title": title, "xaxis": {"title": "cycles"}, "yaxis": {"title": "quality"}})


def content_figure(qc, div, title):
    n = qc.readLen
    xs = list(range(n))
    series = [qc.percents[b][:n].tolist() for b in ALL_BASES] + [qc.gcPercents[:n].tolist()]
    traces = _lines(xs, series, [_BASE_COLOURS[b] for b in ALL_BASES] + ["rgba(20,20,20,255)"], list(ALL_BASES) + ["GC"])
    return Figure(title, div, traces, {"title": title, "xaxis": {"title": "cycles"}, "yaxis": {"title": "percents", "range": [0.0, 0.8]}})


def gc_figure(qc, div, title):
    n = qc.readLen
    if n == 0:
        return Figure(title, div, None, None)
    from . import capi
    # (squeeze(), qualitycontrol.py:59-71, has cut the histogram to readLen entries before the report is drawn —
    #  preprocesser.py:703-708 — so upstream plots readLen + 1 x positions against readLen counts; kept)
    hist = qc.acc[capi.QC_GC_HIST, :n].tolist()
    xs = [100.0 * float(t) / n for t in range(n + 1)]
    return Figure(title, div, [{"x": xs, "y": hist, "type": "bar"}], {"title": title, "xaxis": {"title": "percents(%)"}, "yaxis": {"title": "counts"}})


def discontinuity_figure(qc, div, title):
    n = qc.readLen
    ys = qc.meanDiscontinuity[:n].tolist()
    top = (max(ys) if ys else 0.0) * 1.5
    return Figure(title, div, [{"x": list(range(n)), "y": ys, "mode": "lines", "line": {"color": "rgba(100,150,0,0.5)", "width": 2}}],
                  {"title": title, "xaxis": {"title": "cycles"}, "yaxis": {"title": "discontinuity", "range": [0.0, top]}})


def strand_bias_figure(qc, div, title):
    """forward vs reverse-complement counts of up to 1000 k-mers taken at equal steps from the count-sorted list, the 50
    most frequent skipped (qualitycontrol.py:238-270; its `/` is python-2 integer division)"""
    if qc.readLen == 0:
        return Figure(title, div, None, None)
    total = len(qc.topKmerCount)
    shift = min(50, total // 2)
    top = min(total - shift, 1000)
    step = max(1, (total - shift) // top) if top > 0 else 1
    fwd, rev, hi = [0] * max(top, 0), [0] * max(top, 0), 0
    picked = [i * step + shift for i in range(max(top, 0)) if i * step + shift < total]      # (the reference breaks at the first index >= total)
    if picked:
        f, r = qc.kmer_pairs(picked)
        fwd[:len(f)], rev[:len(r)] = f, r
        hi = max(0, max(f), max(r))
    return Figure(title, div, [{"x": fwd, "y": rev, "mode": "markers", "type": "scatter", "marker": {"size": 2, "color": "rgba(0,0,50,128)"}}],
                  {"title": title, "xaxis": {"title": "relative forward strand KMER count", "range": [-10, hi]},
                   "yaxis": {"title": "relative reverse strand KMER count", "range": [-10, hi]}})


# ---- run-level figures (qualitycontrol.py:272-322, preprocesser.py:677-700) ------------------------------------------------------
def filter_stat_figure(summary, paired, debubble, barcode):
    labels = ["good reads", "has_polyX", "low_quality", "too_short", "too_many_N"]
    counts = [summary["good_reads"], summary["bad_reads_with_polyX"], summary["bad_reads_with_low_quality"],
              summary["bad_reads_with_bad_read_length"], summary["bad_reads_with_too_many_N"]]
    if paired:
        labels.append("bad_overlap"); counts.append(summary["bad_reads_with_bad_overlap"])
    if debubble:
        labels.append("in_bubble"); counts.append(summary["bad_reads_with_reads_in_bubble"])
    if barcode:
        labels.append("bad_barcode"); counts.append(summary["bad_reads_with_bad_barcode"])
    total = summary["total_reads"]
    labels = ["%s: %d(%s%%)" % (l, c, str(100.0 * float(c) / total if total > 0 else 0.0)) for l, c in zip(labels, counts)]
    title = "Filtering statistics of sampled %d reads" % total
    return Figure("Good reads and bad reads after filtering", "filter_stat",
                  [{"values": counts, "labels": labels, "textinfo": "none", "type": "pie"}], {"title": title, "width": 800, "height": 600})


def error_figure(matrix):
    names, values, colours = [], [], []
    transitions = {("A", "G"), ("G", "A"), ("C", "T"), ("T", "C")}
    for a in ALL_BASES:
        for b in ALL_BASES:
            if a != b:
                names.append(a + "->" + b)
                values.append(matrix[a][b])
                colours.append("rgba(246, 103, 0,1.0)" if (a, b) in transitions else "rgba(22, 96, 167,1.0)")
    return Figure("Sequence error distribution", "error_matrix", [{"x": names, "y": values, "marker": {"color": colours}, "type": "bar"}],
                  {"title": "sequencing error transform distribution", "xaxis": {"title": "seq error transform"}, "yaxis": {"title": "counts"}})


def overlap_figure(hist, read_len, total_reads):
    none_pct = int(hist[0] * 100.0 / total_reads) if total_reads > 0 and len(hist) else 0
    return Figure("Overlap length distribution", "overlap_stat", [{"x": list(range(read_len + 1)), "y": list(hist), "type": "bar"}],
                  {"title": "Pair overlap Length Histgram", "xaxis": {"title": "overlap Length (%d%% not overlapped)" % none_pct, "range": [-2, read_len]},
                   "yaxis": {"title": "counts"}})


def qc_figures(qc, tag, div_tag, when):
    """the five figures of one QualityControl object; tag '' / 'Read1 ' / 'Read2 ', when 'before' / 'after'"""
    cap = (lambda s: s) if tag else (lambda s: s[0].upper() + s[1:])
    kmer_word = "Kmer" if when == "before" and tag else "kmer"
    pre = "pre" if when == "before" else "post"
    out = [quality_figure(qc, "%s_%s_quality" % (div_tag, pre), cap(tag + "quality curve %s filtering" % when)),
           content_figure(qc, "%s_%s_content" % (div_tag, pre), cap(tag + "base content distribution %s filtering" % when)),
           gc_figure(qc, "%s_%s_gc" % (div_tag, pre), cap(tag + "GC curve %s filtering" % when)),
           discontinuity_figure(qc, "%s_%s_discontinuity" % (div_tag, pre), cap(tag + "discontinuity curve %s filtering" % when)),
           strand_bias_figure(qc, "%s_%s_sb" % (div_tag, pre), cap(tag + "%s strand bias %s filtering" % (kmer_word, when)))]
    # the menu / section titles differ slightly from the plot titles for two of them (preprocesser.py:790-791)
    out[3].menu = cap(tag + "per base discontinuity %s filtering" % when)
    out[4].menu = cap(tag + "kmer strand bias %s filtering" % when)
    return out


CSS = """<style type="text/css">
#menu {text-align:left;}
.menu-item{font-size:14px;padding:4px;}
#container {text-align:center;padding-left:30px;}
.figure-title {color:#bbbbbb;font-size:30px;padding:10px;text-align:left;}
.figure-div {margin-top:40px;text-align:center}
.summary-table {padding:5px;border:1px solid #eeeeee;width:800px}
.col1 {text-align:right;padding:5px;padding-right:20px;color:#666666;}
.col2 {text-align:left;padding:5px;padding-left:20px;color:#332299;}
.plotly-div {width:800;height:600;text-align:center;}
li {color:#666666;font-size:15px;border:0px;}
</style>
"""


def summary_rows(stat, version):
    s, cmd = stat["afterqc_main_summary"], stat["command"]
    paired = cmd["read2_file"] is not None
    grey = lambda t: " <font color='#aaaaaa'>(" + t + ")</font>"
    pct = lambda a, b: "%0.3f%%" % (100.0 * float(a) / float(b)) if b else "0.000%"
    lost = s["total_bases"] - s["good_bases"]
    rows = [("AfterQC Version:", version),
            ("sequencing:", ("2*%d pair end" if paired else "%d single end") % s["readlen"]),
            ("total reads:", human(s["total_reads"])),
            ("filtered out reads:", "%0.3f" % s["bad_reads"] + grey(pct(s["bad_reads"], s["total_reads"]))),
            ("total bases:", human(s["total_bases"])),
            ("filtered out bases:", "%0.3f" % lost + grey(pct(lost, s["total_bases"])))]
    if paired:
        ov = stat["afterqc_overlap"]
        rows += [("estimated seq error:", "%0.3f%%" % (ov["error_rate"] * 100)),
                 ("adapter trimmed reads:", human(ov["trimmed_adapter_reads"])),
                 ("adapter trimmed bases:", human(ov["trimmed_adapter_bases"]))]
    rows.append(("auto trimming", "front:%s, tail:%s (use <font color='#aaaaaa'>-f0 -t0</font> to disable)" % (cmd["trim_front"], cmd["trim_tail"])))
    return rows


def render(stat, figures, version):
    """the whole page as a string"""
    menu_of = lambda f: getattr(f, "menu", f.title)
    out = ["<HTML>\n<HEAD>\n", '<script src="http://cdn.plot.ly/plotly-latest.min.js"></script>\n', CSS, "</HEAD>\n<BODY>\n<DIV id='container'>\n"]
    out.append("<div id='menu'><ul>\n<li class='menu-item'><a href='#summary'>1, AfterQC summary</a> </li>\n")
    for k, f in enumerate(figures):
        out.append("<li class='menu-item'><a href='#%s'>%d, %s</a> </li>\n" % (div_id(menu_of(f)), k + 2, menu_of(f)))
    out.append("</ul></div>\n<div class='figure-div'>\n<div class='figure-title'><a name='summary'>1, AfterQC summary</a></div>\n<table class='summary-table'>\n")
    for k, v in summary_rows(stat, version):
        out.append("<tr><td class='col1'>%s</td><td class='col2'>%s</td></tr>\n" % (k, v))
    out.append("</table>\n</div>\n<div id='figures'>")
    for k, f in enumerate(figures):
        out.append("<div class='figure-div'>\n<div class='figure-title'><a name='%s'>%d, %s</a></div>\n" % (div_id(menu_of(f)), k + 2, menu_of(f)))
        out.append("<div id='%s' class='plotly-div'></div>\n<div class='figure-summary'></div>\n</div>\n" % f.div)
    out.append("</div>\n</DIV>\n<script type=\"text/javascript\">\n")
    for f in figures:
        out.append(f.script())
        out.append("\n")
    out.append("</script>\n</BODY>\n</HTML>")
    return "".join(out)


def build_figures(stat, opt, r1pre, r2pre, r1post, r2post, overlap_hist, read_len):
    """the figure list in the reference's order (preprocesser.py:700, 771-772, 785-830)"""
    paired = opt.read2_file is not None
    figs = [filter_stat_figure(stat["afterqc_main_summary"], paired, bool(opt.debubble), bool(opt.barcode))]
    if paired:
        figs.append(error_figure(stat["afterqc_overlap"]["error_matrix"]))
        figs.append(overlap_figure([int(v) for v in overlap_hist[:read_len + 1]], read_len, stat["afterqc_main_summary"]["total_reads"]))
        for tag, dv, pre, post in (("Read1 ", "r1", r1pre, r1post), ("Read2 ", "r2", r2pre, r2post)):
            figs += qc_figures(pre, tag, dv, "before") + qc_figures(post, tag, dv, "after")
    else:
        figs += qc_figures(r1pre, "", "r1", "before") + qc_figures(r1post, "", "r1", "after")
    return figs
