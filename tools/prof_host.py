import cProfile, pstats, sys, io
sys.argv=['bench.py','--steps','3','--warmup','1','--no-cpu-baseline']
sys.path.insert(0,'/root/repo')
import runpy
pr=cProfile.Profile()
pr.enable()
try:
    runpy.run_path('/root/repo/bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s=io.StringIO()
ps=pstats.Stats(pr,stream=s).sort_stats('cumulative')
ps.print_stats(45)
print(s.getvalue()[:9000])
