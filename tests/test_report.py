"""CPU: the HTML report (QC/<read1>.html, qcreporter.py + qualitycontrol.py:158-322 upstream) — page skeleton, the
reference's figure list and div ids, and figure data that equals the stats JSON written next to it."""
import json
import os
import re

import gzip

import pytest

from conftest import GOLDEN
from test_host_golden import run_case

PE_DIVS = ["filter_stat", "error_matrix", "overlap_stat"] + ["r%d_%s_%s" % (m, w, k) for m in (1, 2) for w in ("pre", "post")
                                                               for k in ("quality", "content", "gc", "discontinuity", "sb")]
SE_DIVS = ["filter_stat"] + ["r1_%s_%s" % (w, k) for w in ("pre", "post") for k in ("quality", "content", "gc", "discontinuity", "sb")]


def figures_of(html):
    out = {}
    for m in re.finditer(r"var data=(.*?);\nvar layout=(.*?);\nPlotly\.newPlot\('([a-z0-9_]+)', data, layout\);", html, re.S):
        out[m.group(3)] = (json.loads(m.group(1)), json.loads(m.group(2)))
    return out


@pytest.mark.parametrize("name,divs", [("g1_testdata", PE_DIVS), ("se_default", SE_DIVS), ("pe_barcode", PE_DIVS)])
def test_report_matches_stats(name, divs, tmp_path, e2e):
    from oracle import oracle
    work, stat = run_case(name, tmp_path, oracle.OracleEngine(), "text")
    rec = e2e[name]
    html_path = os.path.join(work, rec["stat_file"][:-5] + ".html")
    assert os.path.exists(html_path)
    html = open(html_path).read()
    assert html.startswith("<HTML>") and html.rstrip().endswith("</HTML>") and "plotly-latest.min.js" in html
    figs = figures_of(html)
    assert list(figs) == divs                                          # the reference's order
    for d in divs:
        assert "<div id='%s' class='plotly-div'></div>" % d in html
    # menu and section numbering: summary is 1, figures follow
    assert html.count("class='menu-item'") == len(divs) + 1
    assert "<a name='summary'>1, AfterQC summary</a>" in html
    with open(os.path.join(work, rec["stat_file"])) as f:
        js = json.load(f)
    s = js["afterqc_main_summary"]
    paired = js["command"]["read2_file"] is not None
    # quality curves == the JSON's base_quality / mean_quality, content == base_content / gc_content
    for key, tag in (("read1_prefilter", "r1_pre"), ("read1_postfilter", "r1_post")) + ((("read2_prefilter", "r2_pre"), ("read2_postfilter", "r2_post")) if paired else ()):
        data, layout = figs[tag + "_quality"]
        assert [t["name"] for t in data] == ["A", "T", "C", "G", "mean"]
        for t in data[:4]:
            assert t["y"] == js["base_quality"][key][t["name"]]
        assert data[4]["y"] == js["mean_quality"][key]
        data, layout = figs[tag + "_content"]
        for t in data[:4]:
            assert t["y"] == js["base_content"][key][t["name"]]
        assert data[4]["y"] == js["gc_content"][key] and layout["yaxis"]["range"] == [0.0, 0.8]
        data, _ = figs[tag + "_sb"]
        assert len(data[0]["x"]) == len(data[0]["y"]) <= 1000
    pie, lay = figs["filter_stat"]
    assert pie[0]["values"][0] == s["good_reads"] and lay["title"] == "Filtering statistics of sampled %d reads" % s["total_reads"]
    assert sum(pie[0]["values"]) == s["total_reads"] - (0 if js["command"]["barcode"] or not s["bad_reads_with_bad_barcode"] else s["bad_reads_with_bad_barcode"]) - \
        (0 if js["command"]["debubble"] else s["bad_reads_with_reads_in_bubble"])
    if paired:
        bars, _ = figs["error_matrix"]
        m = js["afterqc_overlap"]["error_matrix"]
        assert dict(zip(bars[0]["x"], bars[0]["y"])) == {a + "->" + b: m[a][b] for a in "ATCG" for b in "ATCG" if a != b}
        hist, _ = figs["overlap_stat"]
        assert len(hist[0]["y"]) == s["readlen"] + 1 and sum(hist[0]["y"]) <= s["total_reads"]
    assert ("2*%d pair end" % s["readlen"] if paired else "%d single end" % s["readlen"]) in html


# ---- pinned to the reference: the report the REAL reference wrote for the same inputs ---------------------------------------
# tests/golden/report_vectors.json.gz holds, per case, what tests/golden/make_golden.py parsed out of the reference's own
# HTML (qcreporter.py:36-138 page, qualitycontrol.py:158-322 figures, preprocesser.py:785-830 figure list): menu entries,
# summary table rows, section order and, per Plotly div, the traces and the layout.  The one python-2-vs-3 delta of that
# code path — `/` on list lengths in strandBiasPlotly — is restored in the generator with ints whose `/` floors.
def parse_own_report(html):
    rep = {"menu": [list(t) for t in re.findall(r"<li class='menu-item'><a href='#([^']*)'>(\d+), (.*?)</a> </li>", html)],
           "summary": [list(t) for t in re.findall(r"<tr><td class='col1'>(.*?)</td><td class='col2'>(.*?)</td></tr>", html)],
           "sections": [list(t) for t in re.findall(r"<div class='figure-title'><a name='([^']*)'>(\d+), (.*?)</a></div>\n<div id='([^']*)' class='plotly-div'></div>", html)],
           "figures": {d: {"data": data, "layout": layout} for d, (data, layout) in figures_of(html).items()}}
    return rep


REPORT_CASES = ["g1_testdata", "se_default", "pe_barcode", "pe_cfg5"]


def check_report_against_reference(name, work, e2e):
    with gzip.open(os.path.join(GOLDEN, "report_vectors.json.gz"), "rt") as f:
        exp = json.load(f)[name]
    rec = e2e[name]
    with open(os.path.join(work, rec["stat_file"][:-5] + ".html")) as f:
        got = parse_own_report(f.read())
    assert got["menu"] == exp["menu"]
    assert got["sections"] == exp["sections"]
    # every row but the version string (the reference prints its own: 0.9.6)
    assert [r for r in got["summary"] if r[0] != "AfterQC Version:"] == [r for r in exp["summary"] if r[0] != "AfterQC Version:"]
    assert [r[0] for r in got["summary"]] == [r[0] for r in exp["summary"]]
    assert set(got["figures"]) == set(exp["figures"])
    for div, fig in exp["figures"].items():
        mine = got["figures"][div]
        assert mine["layout"] == fig["layout"], (name, div, mine["layout"], fig["layout"])
        assert len(mine["data"]) == len(fig["data"]), (name, div)
        for a, b in zip(mine["data"], fig["data"]):
            assert a.keys() == b.keys(), (name, div, sorted(a), sorted(b))
            for k in b:
                assert a[k] == b[k], (name, div, k)


@pytest.mark.parametrize("name", REPORT_CASES)
def test_report_equals_the_reference_report(name, tmp_path, e2e):
    from oracle import oracle
    work, _ = run_case(name, tmp_path, oracle.OracleEngine(), "text")
    check_report_against_reference(name, work, e2e)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["text", "pipe"])
@pytest.mark.parametrize("name", REPORT_CASES)
def test_report_equals_the_reference_report_on_gpu(name, mode, tmp_path, e2e, gpu_engine):
    """the same through the HIP engine: serial chunk loop and whole-input pipe"""
    work, _ = run_case(name, tmp_path, gpu_engine, mode)
    check_report_against_reference(name, work, e2e)
