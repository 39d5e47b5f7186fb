"""Run bench.py against an A/B build of the library: PROBE_LIB=<name> python tools/bench_variant.py [bench.py arguments]
(make -C object-pose-estimation_amd VARIANT=<name> EXTRA=-D...)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
import bench
sys.exit(bench.main())
