import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "tests")
import numpy as np
from afterqc_amd import synth, capi
from oracle import oracle
from test_gpu_parity import default_cfg
d = synth.make_pairs(n=6000, L=150, seed=4242, dirty=True)
batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
cfg = default_cfg(True)
res = []
for eng in (capi.Engine(0, 2), oracle.OracleEngine()):
    eng.set_config(cfg); eng.reset_stats(); eng.upload(0, batch); eng.run(0); res.append(eng.fetch_results(0))
g, o = res
bad = np.flatnonzero((g.view(np.uint8).reshape(-1, 32) != o.view(np.uint8).reshape(-1, 32)).any(axis=1))
print("differing records:", len(bad), "of", len(g))
for i in bad[:25]:
    s1 = d["seq1"][i].tobytes(); s2 = d["seq2"][i].tobytes()
    print(i, "gpu", g[i]["flag"], g[i]["offset"], g[i]["overlap_len"], g[i]["distance"], "| oracle", o[i]["flag"], o[i]["offset"], o[i]["overlap_len"], o[i]["distance"],
          "| poly1", oracle.hasPolyX(s1, 35, 2), "poly2", oracle.hasPolyX(s2, 35, 2))
print(np.bincount(g["flag"], minlength=12), np.bincount(o["flag"], minlength=12))
raw = g.view(np.uint8).reshape(-1, 32)
for i in bad[:12]:
    print(i, "dbg role0 %s role1 %s" % (bin(raw[i, 28]), bin(raw[i, 29])))
