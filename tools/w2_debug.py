"""A/B of the two weight-gradient kernels on the full-size fixture: run with OPE_WGRAD2=0 and =1 (separate processes), compare the gradient vectors."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from conftest import load_golden
    from gpu_util import build_from_fixture, batch_from
    g = load_golden(sys.argv[3])
    dims, buf, policy, trainer = build_from_fixture(g)
    batch = batch_from(buf, g["inds"])
    trainer.train_policy_on_batch(batch)
    torch.cuda.synchronize()
    np.save(sys.argv[2], trainer.grad.cpu().numpy())
    spec = {"agent/" + k: (tuple(sh), off) for k, (sh, off) in policy.q_network.spec().items()}
    if not trainer.vdn:
        spec.update({"mixer/" + k: (tuple(sh), off) for k, (sh, off) in trainer.mixer.spec().items()})
    import json
    json.dump(spec, open(sys.argv[2] + ".json", "w"))
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "qmix_3s5z_b32"
for v in ("0", "1"):
    subprocess.run([sys.executable, __file__, "child", "/tmp/w2g%s.npy" % v, name], env=dict(os.environ, OPE_WGRAD2=v), check=True, stdout=subprocess.DEVNULL)
import json
a, b = np.load("/tmp/w2g0.npy"), np.load("/tmp/w2g1.npy")
spec = json.load(open("/tmp/w2g0.npy.json"))
for k, (sh, off) in spec.items():
    n = int(np.prod(sh))
    x, y = a[off:off + n].reshape(sh), b[off:off + n].reshape(sh)
    d = np.abs(x - y) / max(np.abs(x).max(), 1e-12)
    bad = np.argwhere(d > 1e-4)
    if len(bad):
        rows = sorted(set(int(r[0]) for r in bad))
        cols = sorted(set(int(r[-1]) for r in bad)) if len(sh) > 1 else []
        print(k, sh, "bad", len(bad), "max", float(d.max()), "rows", rows[:40], "cols", cols[:70])
print("compared", len(spec), "tensors")
