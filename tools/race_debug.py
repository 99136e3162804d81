"""Run the graph test's eager/graph legs repeatedly and print per-step losses (race hunting)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import conftest  # noqa: F401  (sys.path)
from tests.test_gpu_graph import _run

keys = ["img_D/fake", "st_D/fake", "seg_D/fake", "G/loss"]
for tag, g in (("eager", False), ("eager", False), ("graph", True), ("graph", True), ("eager", False), ("graph", True)):
    h, w, used, bn = _run(g)
    print(tag, used)
    for k in keys:
        print("   %-12s" % k, " ".join("%.6f" % s[k] for s in h))
