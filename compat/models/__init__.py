"""Import-path shim: put ``<repo>/compat`` (and ``<repo>``) on sys.path and the reference's dotted
class names (configs/inference_dit.yml:32 ``models.vae3d_dib.VAE``, :53 ``models.dit_crossattn.DiT``,
inference.py:21 ``from models.diffusion import create_diffusion``) resolve to the MI355X path."""
