"""CPU simulation (authoring container): what 16-bit operand rounding does to the teacher's normalised CAMs and label maps.
Fake-quantises every MFMA operand of the fused teacher path (LN output, weights, qkv, P, attention output, fc1 output) to a dtype
while keeping fp32 accumulation / residual / CAM heads, and compares with the fp32 oracle on the same weights and inputs."""
import sys, types
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import torch_oracle as to, c_oracle
from cosa_amd.train_step import synthetic_batch

def make(dt, split=None):
    q = (lambda t: t) if dt is None else (lambda t: t.to(dt).float())
    def encoder(self, x):
        p = self.p
        t, h, w = self.tokens(q(x)) if dt is None else tokens_q(self, x)
        B, N, D = t.shape
        hd = D // self.heads
        embeds = []
        for i in range(self.depth):
            pre = f"encoder.blocks.{i}."
            y = q(F.layer_norm(t, (D,), q(p(pre + "norm1.weight")), q(p(pre + "norm1.bias")), 1e-6))
            qkv = q(F.linear(y, q(p(pre + "attn.qkv.weight")), q(p(pre + "attn.qkv.bias")))).reshape(B, N, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
            att = (qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5
            m = att.amax(-1, keepdim=True)
            e = torch.exp(att - m)
            y = q((q(e) @ qkv[2]) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, N, D)
            t = t + F.linear(y, q(p(pre + "attn.proj.weight")), q(p(pre + "attn.proj.bias")))
            y = q(F.layer_norm(t, (D,), q(p(pre + "norm2.weight")), q(p(pre + "norm2.bias")), 1e-6))
            y = q(F.gelu(F.linear(y, q(p(pre + "mlp.fc1.weight")), q(p(pre + "mlp.fc1.bias")))))
            t = t + F.linear(y, q(p(pre + "mlp.fc2.weight")), q(p(pre + "mlp.fc2.bias")))
            embeds.append(t)
        final = F.layer_norm(t, (D,), q(p("encoder.norm.weight")), q(p("encoder.norm.bias")), 1e-6)
        embeds[-1] = final
        return final[:, 0], final[:, 1:], embeds[self.aux_layer][:, 1:], h, w
    def tokens_q(self, x):
        p = self.p
        B, _, H, W = x.shape
        h, w = H // self.patch, W // self.patch
        t = q(F.conv2d(q(x), q(p("encoder.patch_embed.proj.weight")), q(p("encoder.patch_embed.proj.bias")), stride=self.patch))
        t = t.flatten(2).transpose(1, 2)
        pe = p("encoder.pos_embed")
        grid = pe[:, 1:].reshape(1, self.grid, self.grid, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(h, w), mode="bicubic", align_corners=False).reshape(1, -1, h * w).permute(0, 2, 1)
        t = torch.cat([q(p("encoder.cls_token")).expand(B, -1, -1), t], 1)
        return q(t + q(torch.cat([pe[:, :1], grid], 1))), h, w
    return encoder

def miou(a, b, n=21):
    ious = []
    for c in list(range(n)) + [255]:
        A, B = a == c, b == c
        u = (A | B).sum()
        if u: ious.append((A & B).sum() / u)
    return float(np.mean(ious))

S = int(sys.argv[1]) if len(sys.argv) > 1 else 224
torch.manual_seed(3)
from cosa_amd.models import build_model
from cosa_amd.train_step import default_args
args = default_args("VOC12", crop_size=S, compute_dtype=torch.float32)
net = build_model(args)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
wimg, simg, lab, box = synthetic_batch(2, S, 20, torch.device('cpu'), seed=5)
res = {}
for name, dt in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
    m = to.OracleViT(num_classes=21, aux_layer=-4)
    m.load_named(sd)
    m.encoder = types.MethodType(make(dt), m)
    with torch.no_grad():
        cam, cam_aux, seg = to.multi_scale_camseg(m, wimg, [1.0, 0.5, 1.5])
    masks = [c_oracle.cam2mask(None, np.asarray(box.numpy(), np.int32), c.numpy(), lab.numpy(), 0.7, 0.25, 2, par=None) for c in (cam, cam_aux)]
    res[name] = (cam, cam_aux, masks)
    if name != "fp32":
        r = res["fp32"]
        act = lab.bool()
        for k, nm in ((0, "cam"), (1, "cam_aux")):
            rel = ((res[name][k] - r[k]).abs().amax(dim=(2, 3)) / r[k].abs().amax(dim=(2, 3)).clamp_min(1e-6))[act].max().item()
            agree = np.mean(masks[k] == r[2][k]); iou = miou(masks[k], r[2][k])
            print(f"{name} {nm}: rel err {rel:.3e}  label agreement {agree:.5f}  mIoU {iou:.5f}", flush=True)
