"""eager vs hipGraph-replayed step: which parameters / buffers diverge (debug aid for stream races)."""
import os, sys, importlib.util, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safetensors.torch import load_file
spec = importlib.util.spec_from_file_location('_ssl_t', os.path.join(ROOT, 'tests', 'test_ssl_gpu.py'))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
from oracle.make_golden_ssl import SSL_CFG as C
from vtp_amd import VTPTrainer
g = load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny_ssl.safetensors"))
sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
DEV = "cuda"
img = torch.randn(C["B"], 3, C["R"], C["R"], device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
txt = torch.randint(1, 60, (C["B"], 8), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4)); txt[:, 5] = 63
use_txt = os.environ.get("PROBE_TXT", "1") == "1"
nsteps = int(os.environ.get("PROBE_STEPS", "2"))
def run(use_graphs):
    torch.manual_seed(0)
    m = mod.build_vtp(sd)
    tr = VTPTrainer(m, lr=5e-4, weight_decay=0.0, use_graphs=use_graphs)
    ssl = tr.prepare_ssl(g["in.global_crops"].to(DEV), g["in.local_crops"].to(DEV), g["in.masks"].bool())
    snaps = []
    for _ in range(nsteps):
        tr.step(img, txt if use_txt else None, ssl)
        torch.cuda.synchronize()
        st = m._engine()
        snaps.append((st.flat_p.clone(), st.flat_g.clone(), tr.center_dino.clone(), tr.center_ibot.clone(), float(tr.ssl_loss_sum)))
    return m, snaps
for trial in range(int(os.environ.get("PROBE_TRIALS", "3"))):
    m, a = run(False)
    _, b = run(True)
    st = m._engine()
    for s in range(nsteps):
        bad = []
        for n, (o, k) in st.offsets.items():
            for which, nm in ((1, "g"), (0, "p")):
                x, y = a[s][which][o:o + k], b[s][which][o:o + k]
                e = float((x - y).norm() / (x.norm() + 1e-30))
                if e > 1e-3:
                    bad.append(f"{nm}:{n}={e:.1e}")
        ec = [float((a[s][i] - b[s][i]).norm() / (a[s][i].norm() + 1e-30)) for i in (2, 3)]
        print(f"trial {trial} step {s}: loss {a[s][4]:.5f} vs {b[s][4]:.5f} centres {ec[0]:.1e} {ec[1]:.1e} bad({len(bad)}): {' '.join(bad[:12])}")
