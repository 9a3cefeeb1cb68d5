"""`bench.py` with the session classes answered by the CPU oracle (tests/dryrun_plugin.py): exercises the host side of
the benchmark -- worker threads, helper processes, the per-step gather across ranks, the single JSON line -- on a
machine without a GPU.  Test infrastructure; the numbers it prints mean nothing."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import dryrun_plugin  # noqa: E402

dryrun_plugin.pytest_configure(None)
import bench  # noqa: E402

bench.main()
