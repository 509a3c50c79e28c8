import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
from phase_timing import run
run('nao',3072, solver_iters=0)
run('nao',3072, solver_iters=0, contact_margin=-1.0)
run('nao',3072, solver_iters=0, max_contacts=1)
run('nao',3072, solver_iters=0, max_contacts=2)
run('nao',3072, solver_iters=0, max_contacts=4)
run('nao',3072, solver_iters=0, max_contacts=8)
run('nao',3072, solver_iters=0, enable_ground=False)
