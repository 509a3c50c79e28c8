import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
from phase_timing import run
for n in (1024, 2048, 3072, 4096, 6144):
    run('nao', n, solver_iters=0)
    run('nao', n, solver_iters=50)
    run('nao', n, solver_iters=200)
