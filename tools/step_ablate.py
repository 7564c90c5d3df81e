"""Whole-step ablation: run bench.py's step with aas_set_debug_flags(FLAGS) (e.g. 128 = GEMMs reduced to their epilogue)
to see how much of the step's critical path a kernel class holds.  Numerics are garbage under ablation flags."""
import sys

sys.path.insert(0, ".")
from aas_enhancement_amd import _lib  # noqa: E402

flags = int(sys.argv[1])
_lib.lib().aas_set_debug_flags(flags)
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()
