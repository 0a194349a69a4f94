"""Round 4: what each activation-storage policy costs, measured with the CPU oracle (fp32 arithmetic, boundary tensors rounded per policy; inputs rounded to
bf16 as the engine receives them).  Columns: error against the fp32 truth on fp32 inputs | against the truth on the same bf16-rounded inputs.
    python profiles/scripts/storage_study.py en_s0 > profiles/r04_storage_study.txt   (needs tests/golden fixtures; CPU only)"""
import sys, torch, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from golden_util import Golden, ctor_kwargs
from achelous_amd import Achelous
from achelous_amd.synth import condition_state_dict, make_inputs
from oracle.achelous_oracle import AchelousOracle
OUT = ('det0','det1','det2','se_seg','lane_seg','pc_seg')
torch.set_num_threads(8)
def rel(a,b): return ((a.double()-b.double()).abs().max()/(b.double().abs().max()+1e-6)).item()
class Pol(AchelousOracle):
    def __init__(self, *a, policy=None, **k):
        super().__init__(*a, **k); self.policy = policy; self.boundary_dtype = torch.bfloat16  # so inputs get rounded
    def _b(self, name, t):
        d = self.policy(name, t)
        if d is not None: t = t.to(d).float()
        self.taps[name] = t
        return t
def mk(thr, small, big=torch.bfloat16):
    def p(name, t):
        hw = t.shape[-1]*t.shape[-2] if t.dim()==4 else 10**9
        return small if hw <= thr else big
    return p
name = sys.argv[1]
g = Golden(name); kw = ctor_kwargs(g.meta)
m = Achelous(**kw).eval()
m.load_state_dict(g.calibrate(condition_state_dict(m.state_dict(), seed=g.meta['weight_seed'])), strict=True)
sd = {k: v.cpu() for k, v in m.state_dict().items()}
x, xr, xp = make_inputs(g.meta['batch'], g.meta['input_seed'], resolution=kw['resolution'], pc_channels=kw['pc_channels'])
okw = {k: kw[k] for k in ('num_det','num_seg','phi','backbone','neck','pc_seg','pc_channels','pc_classes','nano_head','spp','resolution')}
t = AchelousOracle(sd, **okw).forward(x, xr, xp)
truth = dict(zip(OUT, (*t[0], t[1], t[2], t[3])))
t2 = AchelousOracle(sd, **okw).forward(x.bfloat16().float(), xr.bfloat16().float(), xp.bfloat16().float())
truth2 = dict(zip(OUT, (*t2[0], t2[1], t2[2], t2[3])))
print('input rounding alone:', {k: f'{rel(truth2[k], truth[k]):.1e}' for k in OUT})
pols = {'all bf16': mk(0, None), 'f16<=40^2': mk(1600, torch.float16), 'f16<=80^2': mk(6400, torch.float16), 'f16<=160^2': mk(25600, torch.float16),
        'all f16': mk(10**9, torch.float16), 'f32<=40^2': mk(1600, None), 'f32<=80^2': mk(6400, None)}
for pn, pol in pols.items():
    o = Pol(sd, **okw, policy=pol)
    r = o.forward(x, xr, xp.bfloat16().float())
    got = dict(zip(OUT, (*r[0], r[1], r[2], r[3])))
    am_se = (got['se_seg'].argmax(1)==truth['se_seg'].argmax(1)).float().mean().item()
    am_ln = (got['lane_seg'].argmax(1)==truth['lane_seg'].argmax(1)).float().mean().item()
    print(f'{name} {pn:12s}', {k: f'{rel(got[k], truth[k]):.1e}|{rel(got[k], truth2[k]):.1e}' for k in OUT}, f'argmax se {am_se:.4f} lane {am_ln:.4f}', flush=True)
