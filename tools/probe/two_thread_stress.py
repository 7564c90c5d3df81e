"""Stress of tests/test_gpu_round6.py::test_two_host_threads_two_trainers_bit_equal_to_each_alone: the test body N times in one process."""
import sys
import traceback

import torch

sys.path.insert(0, ".")
from tests import test_gpu_round6 as t6  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for i in range(n):
    try:
        t6.test_two_host_threads_two_trainers_bit_equal_to_each_alone(torch.device("cuda:0"))
    except AssertionError as e:
        bad += 1
        msg = traceback.format_exc()
        print("run %d FAILED: %s" % (i, msg[-1500:]), flush=True)
print("%d of %d runs failed" % (bad, n))
