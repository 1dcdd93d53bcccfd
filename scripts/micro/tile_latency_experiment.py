import sys, os, json, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, bench
sys.argv=[sys.argv[0],"--no-extras","--no-cpu-baseline","--refine-iters","0"]
args=bench.parse()
B=bench.Bench(args)
loop,cams=B.build("fused",0.0)
B.run_steps(loop,60)
r=B.profiled(loop,40,1<<bench.PK_FUSED,True)
el,_=B.timed(loop,100)
print(json.dumps({"SGR_DEBUG":os.environ.get("SGR_DEBUG","0"),"fused_ms":round(r[bench.PK_FUSED][0],5),"ms_per_step":round(1e3*el/100,4)}))
