"""Why no order-free certificate can cover the benchmark's images: the one record
between components that the reference merges in its second phase, with the object sizes at which the
priority that put it into the queue was computed.

Builds an INSTRUMENTED copy of oracle/csegment_oracle.cpp in /tmp (two arrays: the object sizes at a
record's last re-score; a line per pop of a record between two objects of >= 64 pixels) and runs it on
one golden image (CPU, ~90 s at 512x1024).  Test infrastructure only.

    python tests/tools/stale_priority_probe.py [golden name]    (default cseg_synth_512x1024_s1000)
"""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

src = open(os.path.join(ROOT, "oracle", "csegment_oracle.cpp")).read()
src = src.replace("std::vector<int> rmcls;",
                  "std::vector<int> rmcls;\n  std::vector<int> rn1, rn2;   // object sizes when the record was last scored")
src = src.replace("""    rprio[r] = (roml[r] * omf + rcdl[r]) / den + bias;""",
                  """    rprio[r] = (roml[r] * omf + rcdl[r]) / den + bias;
    if ((int)rn1.size() <= r) { rn1.resize(r1.capacity() + 1, 0); rn2.resize(r1.capacity() + 1, 0); }
    rn1[r] = osize[a]; rn2[r] = osize[b];""")
src = src.replace("""      n_live_pops++;
      rescore(r);""", """      n_live_pops++;
      const bool big = osize[r1[r]] >= 64 && osize[r2[r]] >= 64;
      const int sn1 = rn1[r], sn2 = rn2[r];
      rescore(r);
      if (big && roml[r] < 0)
        fprintf(stderr, "pop record %d objects (%d,%d) sizes now (%d,%d) queued priority scored at sizes (%d,%d) "
                "log-odds %.3f class delta %.3f queued %.6g fresh %.6g -> %s\\n",
                r, r1[r], r2[r], osize[r1[r]], osize[r2[r]], sn1, sn2, roml[r], rcdl[r], q, rprio[r],
                rprio[r] == q ? "MERGE" : (rprio[r] >= 0 ? "requeue" : "drop"));""")
assert "rn1[r] = osize[a]" in src and "queued priority scored" in src

tmp = tempfile.mkdtemp(prefix="stale_probe_")
cpp, so = os.path.join(tmp, "probe.cpp"), os.path.join(tmp, "libprobe.so")
open(cpp, "w").write(src)
subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", cpp, "-o", so], check=True)

import golden_util as gu          # noqa: E402
from oracle import checker as ck  # noqa: E402

lib = ctypes.CDLL(so)
orig = ck._load
ck._load = lambda name: lib if "csegment_oracle" in name else orig(name)
name = sys.argv[1] if len(sys.argv) > 1 else "cseg_synth_512x1024_s1000"
g = gu.load(name)
sdb, omf, bias = g["spec"]["opts"]
print("image %s, options %s" % (name, g["spec"]["opts"]))
r = ck.run_csegment(g["class_probs"], g["sameness_probs"], g["spec"]["C"], g["offsets"], sdb, omf, bias)
print("objects %d, instances %d" % (r.stats["n_objects"], len(r.object_class)))
print("For each MERGE line above: the record was in the queue because its last re-score -- at the sizes shown --")
print("gave a priority >= 0; (log-odds + class delta) / n + bias >= 0 needs n >= -(log-odds + class delta) / bias.")
