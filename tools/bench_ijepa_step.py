"""End-to-end I-JEPA pretraining step at BASELINE configs[4] scale on one GPU: ViT-L/16 context encoder + EMA target
encoder + 12-block predictor (384 wide), 224^2 images, 4 target blocks, bf16 autocast, AdamW.

The ViT below is a plain timm-style pre-LN stack with mmlearn's module layout (mmlearn/modules/layers/
{attention,mlp,transformer_block}.py names: norm1 / attn.qkv / attn.proj / norm2 / mlp = Sequential(fc1, GELU, Dropout,
fc2, Dropout)), written here so the tool is self-contained; the task, masks, target / context / predictor plumbing and
loss are mmlearn_amd's (rows B1-B7).  Prints stock-blocks vs ``fused.accelerate_encoder`` timings.
    python tools/bench_ijepa_step.py [--batch 128] [--steps 6] [--small]
"""
import argparse, json, os, sys, time
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import ops
from mmlearn_amd.tasks import IJEPA


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.attn_drop, self.proj_drop = nn.Dropout(0.0), nn.Dropout(0.0)

    def forward(self, x):
        B, L, E = x.shape
        q, k, v = self.qkv(x).view(B, L, 3, self.num_heads, E // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
        return self.proj_drop(self.proj(o.transpose(1, 2).reshape(B, L, E))), None


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn, self.drop_path = Attention(dim, num_heads), nn.Identity()
        hid = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hid), nn.GELU(), nn.Dropout(0.0), nn.Linear(hid, dim), nn.Dropout(0.0))

    def forward(self, x, return_attention=False):
        x = x + self.drop_path(self.attn(self.norm1(x))[0])
        return x + self.drop_path(self.mlp(self.norm2(x)))


class PatchEmbed(nn.Module):
    def __init__(self, dim, img=224, patch=16):
        super().__init__()
        self.num_patches = (img // patch) ** 2
        self.proj = nn.Conv2d(3, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class ViT(nn.Module):
    def __init__(self, dim, depth, heads):
        super().__init__()
        self.embed_dim, self.num_heads = dim, heads
        self.patch_embed = PatchEmbed(dim)
        self.pos_embed = nn.Parameter(0.02 * torch.randn(1, self.patch_embed.num_patches, dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, inputs):
        x = self.patch_embed(inputs["rgb"]) + self.pos_embed
        masks = inputs.get("rgb_mask")
        if masks is not None:
            x = ops.apply_masks(x, masks if isinstance(masks, list) else [masks])
        for blk in self.blocks:
            x = blk(x)
        return (self.norm(x), None)


class Predictor(nn.Module):
    """Reference parameter names (mmlearn/modules/encoders/vision.py:441-569); the front-end (token assembly) is
    mmlearn_amd.predictor's, which recognises this layout."""

    def __init__(self, num_patches, embed_dim, pred_dim, depth, heads):
        super().__init__()
        self.num_patches, self.embed_dim, self.num_heads = num_patches, embed_dim, heads
        self.predictor_embed = nn.Linear(embed_dim, pred_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, pred_dim))
        self.predictor_pos_embed = nn.Parameter(0.02 * torch.randn(1, num_patches, pred_dim), requires_grad=False)
        self.predictor_blocks = nn.ModuleList([Block(pred_dim, heads) for _ in range(depth)])
        self.predictor_norm = nn.LayerNorm(pred_dim, eps=1e-6)
        self.predictor_proj = nn.Linear(pred_dim, embed_dim)


def build(small, fused: bool, dev, capturable: bool = False, true_ema: bool = False):
    """``small``: False = ViT-L/16 + 12 x 384 predictor (configs[4]), True = a two-block toy, "vits" = ViT-S/16 + 6 x 384 predictor
    (projects/ijepa/configs/experiment/in1k_vit_small.yaml:52-57, with 64-wide predictor heads)."""
    torch.manual_seed(0)
    if small == "vits":
        dim, depth, heads, pdim, pdepth, pheads = 384, 12, 6, 384, 6, 6
    elif small:
        dim, depth, heads, pdim, pdepth, pheads = 256, 2, 4, 128, 2, 2
    else:
        dim, depth, heads, pdim, pdepth, pheads = 1024, 24, 16, 384, 12, 6
    enc = ViT(dim, depth, heads)
    pred = Predictor(196, dim, pdim, pdepth, pheads)   # 64-wide predictor heads
    if fused:
        from mmlearn_amd.fused import accelerate_encoder
        accelerate_encoder(enc, fuse_qkv=True, fuse_add_ln=True)    # norm1 / norm2 of the pre-LN blocks emit bf16 (automatic)
        accelerate_encoder(pred, fuse_qkv=True, fuse_add_ln=True)
        from mmlearn_amd.optim import AdamW
        optimizer = partial(AdamW, lr=1e-4, weight_decay=0.05, capturable=capturable)
    else:
        optimizer = partial(torch.optim.AdamW, lr=1e-4, weight_decay=0.05, capturable=capturable)
    task = IJEPA(encoder=enc, predictor=pred, optimizer=optimizer, true_ema=true_ema).to(dev)
    task.configure_model()
    return task


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--only", choices=("stock", "fused"), default=None, help="run one variant (for profiling)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    imgs = torch.rand(args.batch, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
    out = {"batch": args.batch, "model": "small" if args.small else "ViT-L/16 + 12x384 predictor"}
    for fused in ((False, True) if args.only is None else (args.only == "fused",)):
        task = build(args.small, fused, dev)
        opt = task.configure_optimizers()
        opt = opt["optimizer"] if isinstance(opt, dict) else opt

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step({"rgb": imgs}, 0)
            loss.backward()
            opt.step()
            task.on_before_zero_grad(opt)   # EMA update of the target encoder
            return loss

        torch.manual_seed(7)
        for _ in range(5):   # MIOpen picks its patch-embedding convolution over the first calls of the stock variant
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        key = "fused" if fused else "stock_blocks"
        out[key + "_ms"] = round(ms, 1)
        out[key + "_images_per_s"] = round(args.batch / ms * 1e3, 1)
        out[key + "_loss"] = round(float(loss.detach().float()), 4)
        del task, opt
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
