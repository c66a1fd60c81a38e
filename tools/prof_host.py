import cProfile, pstats, sys, io, os
sys.argv = ["bench.py", "--config", "c5", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
open("gpurun_out/cprof_c5.txt", "w").write(s.getvalue())
