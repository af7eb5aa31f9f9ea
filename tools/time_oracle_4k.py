import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, resource
from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import synthetic
for nt in (32, 64):
    torch.set_num_threads(nt)
    H, W = 2160, 3840
    weights = synthetic.synthetic_conv_weights(0)
    oracle = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), [0, 5, 10, 19, 28], [21])
    content = synthetic.synthetic_image(0, H, W); style = synthetic.synthetic_image(1, 512, 512)
    t0 = time.time(); oracle.set_targets(style, content); t1 = time.time()
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(0))
    s, c, t, g = ocm.loss_and_grad(oracle, x, 1e5, 1.0)
    t2 = time.time()
    print(f"threads {nt}: set_targets {t1 - t0:.1f} s, loss_and_grad {t2 - t1:.1f} s, total {float(t):.6f}, peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB", flush=True)
