"""Worker of tests/test_gpu_parity.py::test_split_weight_gradients_stay_close_to_fp32_mfma: one REINFORCE iteration (no
optimiser step) of a fixed case, engine gradients saved to OUT.  The library reads JN_WW_EXACT once per process, hence a
process per mode.   usage: split_grad_worker.py OUT"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch


def main():
    import jolineedle_amd as ja
    from tests.helpers import make_pair, synth_batch
    P, Tn, B = 96, 4, 3
    product, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, max_batch=B)
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=61)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(8))
    cfg = ja.CfgNode(max_seq_len=Tn, entropy_weight=0.01, stop_enabled=True, reward_norm=True, seed=0, learning_rate=1e-3,
                     gradient_accumulation=1)
    tr = ja.ReinforceTrainer(cfg, product)
    tr.last_return_mean, tr.last_return_std = 0.25, 1.5
    env = ja.NeedleGeneralEnv(images.cuda(), bboxes, P, Tn, 1, True)
    tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
    torch.save({k: v.detach().cpu() for k, v in product.engine_grads().items()}, sys.argv[1])


if __name__ == "__main__":
    main()
