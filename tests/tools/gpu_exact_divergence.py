"""First divergence between the exact engine's merge sequence and the oracle's (GPU box, diagnostic).

    python tests/tools/gpu_exact_divergence.py family seed      (families of gpu_exact_campaign.py)
Builds an instrumented copy of oracle/csegment_oracle.cpp in a temp dir (one fwrite per merge), runs
both on the same input and prints the first merge at which (survivor, absorbed, priority bits) differ.
"""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np

from gpu_exact_campaign import make

family, seed = sys.argv[1], int(sys.argv[2])
tmp = tempfile.mkdtemp(prefix="xdiv_")
src = open(os.path.join(ROOT, "oracle", "csegment_oracle.cpp")).read()
src = src.replace("""    if (osize[a] < osize[b]) std::swap(a, b);   // a survives; tie keeps the lower id (r1)""",
                  """    if (osize[a] < osize[b]) std::swap(a, b);   // a survives; tie keeps the lower id (r1)
    if (mlogf) { int rec[4] = {a, b, r, 0}; float pp = rprio[r]; memcpy(&rec[3], &pp, 4); fwrite(rec, 16, 1, mlogf); }""")
src = src.replace("long long n_pops = 0,",
                  'FILE* mlogf = getenv("ORACLE_MERGELOG") ? fopen(getenv("ORACLE_MERGELOG"), "wb") : nullptr;\n  long long n_pops = 0,')
src = src.replace("#include <vector>", "#include <vector>\n#include <cstdlib>\n#include <cstring>")
src = src.replace("  s.emit(output, object_class, partition);", "  if (s.mlogf) fclose(s.mlogf);\n  s.emit(output, object_class, partition);")
assert "mlogf" in src
open(os.path.join(tmp, "o.cpp"), "w").write(src)
subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(tmp, "o.cpp"), "-o", os.path.join(tmp, "libo.so")], check=True)

from oracle import checker as ck                 # noqa: E402
from mergenet_amd import segmenter as seg        # noqa: E402

lib = ctypes.CDLL(os.path.join(tmp, "libo.so"))
orig = ck._load
ck._load = lambda name: lib if "csegment_oracle" in name else orig(name)
s, offs, opts = make(family, seed)
C, H, W = s.class_probs.shape
os.environ["ORACLE_MERGELOG"] = os.path.join(tmp, "oracle.bin")
ref = ck.run_csegment(s.class_probs, s.sameness_probs, C, offs, *opts)
os.environ["MN_X_MERGELOG"] = os.path.join(tmp, "engine.bin")
ctx = seg.HostContext(H, W, C, len(offs))
o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1], merge_logprob_bias=opts[2],
                        mode=seg.MN_MODE_EXACT, clip_inputs=1)
mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
a = np.fromfile(os.path.join(tmp, "oracle.bin"), dtype=np.int32).reshape(-1, 4)
b = np.fromfile(os.path.join(tmp, "engine.bin"), dtype=np.int32).reshape(-1, 4)
print("merges: oracle %d engine %d; live pops: oracle %d engine %d; equal partition %s" % (
    len(a), len(b), ref.stats["n_live_pops"], st["finisher_steps"], ck.same_partition(part, ref.partition)))
n = min(len(a), len(b))
same = (a[:n, 0] == b[:n, 0]) & (a[:n, 1] == b[:n, 1]) & (a[:n, 3] == b[:n, 3])
if same.all():
    print("identical merge sequences (%d merges)" % n)
    sys.exit(0)
i = int(np.argmin(same))
print("first divergence at merge %d of %d" % (i, n))
for j in range(max(0, i - 3), min(n, i + 4)):
    pa, pb = a[j, 3:4].view(np.float32)[0], b[j, 3:4].view(np.float32)[0]
    print("%s %7d  oracle (%6d <- %6d) prio %.9g [%08x]   engine (%6d <- %6d) prio %.9g [%08x]" % (
        "->" if j == i else "  ", j, a[j, 0], a[j, 1], pa, int(a[j, 3]) & 0xFFFFFFFF, b[j, 0], b[j, 1], pb, int(b[j, 3]) & 0xFFFFFFFF))
# does the engine's merge appear later in the oracle (a swap of equal priorities) or never?
key = (int(b[i, 0]), int(b[i, 1]))
later = np.nonzero((a[:, 0] == key[0]) & (a[:, 1] == key[1]))[0]
print("the engine's merge %s appears in the oracle's sequence at %s" % (key, later[:3].tolist()))
key = (int(a[i, 0]), int(a[i, 1]))
later = np.nonzero((b[:, 0] == key[0]) & (b[:, 1] == key[1]))[0]
print("the oracle's merge %s appears in the engine's sequence at %s" % (key, later[:3].tolist()))
