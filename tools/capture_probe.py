import sys, os, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = {
"conv": '''
f = tpa.lib.KSpaceFilter(cell, (16,16,16), tpa.CoulombPotential(smearing=1.0))
mesh = torch.randn(1,16,16,16, device=dev, dtype=dt)
def body(): return f(mesh)
''',
"jets": '''
from torchpme_amd import analytic, ops
geom = ops.MeshGeometry(cell.cpu().numpy(), (16,16,16), tpa._lib.P3M, 4)
u = torch.rand(50,3,device=dev,dtype=dt)*16; x = torch.randn(50,1,device=dev,dtype=dt)
def body():
    m = analytic._Spread.apply(u, x, geom, (0,0,0)); return analytic._Gather.apply(u, m, geom, (1,0,0))
''',
"forward": '''
calc.double_backward = "analytic"
def body():
    d = tpa.pair_distances(pos, pairs, cell, S); return calc(q, cell, pos, pairs, d)
''',
"forward_grad": '''
calc.double_backward = "analytic"
def body():
    d = tpa.pair_distances(pos, pairs, cell, S); V = calc(q, cell, pos, pairs, d)
    return torch.autograd.grad((q*V).sum(), pos)[0]
''',
"force_loss": '''
calc.double_backward = "analytic"
theta = torch.ones((), device=dev, dtype=dt, requires_grad=True)
def body():
    qq = q * theta
    d = tpa.pair_distances(pos, pairs, cell, S); V = calc(qq, cell, pos, pairs, d)
    (g,) = torch.autograd.grad((qq*V).sum(), pos, create_graph=True)
    return torch.autograd.grad((g*g).sum(), theta)[0]
''',
"force_loss_auto": '''
calc.double_backward = "auto"
theta = torch.ones((), device=dev, dtype=dt, requires_grad=True)
def body():
    qq = q * theta
    d = tpa.pair_distances(pos, pairs, cell, S); V = calc(qq, cell, pos, pairs, d)
    (g,) = torch.autograd.grad((qq*V).sum(), pos, create_graph=True)
    return torch.autograd.grad((g*g).sum(), theta)[0]
''',
"fused_forward_grad": '''
def body():
    d = tpa.pair_distances(pos, pairs, cell, S); V = calc(q, cell, pos, pairs, d)
    return torch.autograd.grad((q*V).sum(), pos)[0]
''',
}
PRE = '''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import torchpme_amd as tpa
dev = torch.device("cuda",0); dt = torch.float64
rng = np.random.default_rng(0)
n=int(__import__('os').environ.get('PROBE_N','4')); a=2.5
gr=(np.arange(n)+0.5)*a
p=np.stack(np.meshgrid(gr,gr,gr,indexing="ij"),-1).reshape(-1,3)+rng.uniform(-.2,.2,(n**3,3))
c=np.eye(3)*n*a
pr,Sn,_=tpa.neighbor_list(p,c,4.0)
pos=torch.tensor(p,device=dev,requires_grad=True); cell=torch.tensor(c,device=dev); q=torch.tensor(rng.normal(size=(n**3,1)),device=dev)
pairs=torch.tensor(pr,device=dev); S=torch.tensor(Sn,device=dev,dtype=dt)
calc=tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0),mesh_spacing=0.7,interpolation_nodes=4)
''' % ROOT
POST = '''
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): ref = body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print("CAPTURE_OK", float((out-ref).abs().max()))
'''
only = os.environ.get("PROBE_ONLY")
for name, code in PROBES.items():
    if only and name not in only.split(","):
        continue
    r = subprocess.run(["timeout","90",sys.executable,"-c",PRE+code+POST],capture_output=True,text=True)
    tail = (r.stdout+r.stderr).strip().splitlines()
    msg = [l[:160] for l in tail if "CAPTURE_OK" in l or "Error" in l or "error" in l or "dumped" in l or "Segmentation" in l][-3:]
    print(name, "rc", r.returncode, msg)
