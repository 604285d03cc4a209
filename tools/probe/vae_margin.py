import sys, torch, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
g.build()
import topia_xl_amd as pkg
from oracle import synth, vae_ref
from tests.golden.make_golden import SEED, VAE_CFG
gold = np.load("tests/golden/vae_decode.npz")
for dt in (torch.float16, torch.bfloat16):
    vae = pkg.VAE(**VAE_CFG).eval(); sd = synth.state_dict_like(SEED, vae.state_dict()); vae.load_state_dict(sd); vae.to("cuda:0"); vae.compute_dtype = dt
    z = synth.tensor(SEED, "vae.z", (3, 1, 4, 4, 4))
    out = vae.decode(z.cuda()).cpu(); ref = torch.as_tensor(gold["decoded"])
    print(dt, "max_abs", float((out-ref).abs().max()), "rel", float((out-ref).norm()/ref.norm()))
g.smoke()
