import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import ssl4polyp_amd as A
from oracle import vit_mae_ref as O
from oracle import vit_bf16_sim as S
from ssl4polyp_amd.models import _EncoderFrontMixin
from ssl4polyp_amd.engine import BlockStack
cfg = O.VIT_BASE
sd = O.generated_state_dict(cfg, 103, decoder=False, n_class=2)
imgs, labels, _ = O.generated_batch(cfg, 2, 202)
vm = A.get_MAE_backbone(None, True, 2, False, None, precision="bf16")
sdm = dict(sd); sdm["decoder_pos_embed"] = vm.state_dict()["decoder_pos_embed"]
vm.load_state_dict(sdm); vm.cuda()
rt = vm._rt; rt.ensure(torch.device("cuda", 0))
with torch.no_grad():
    cols, x0 = _EncoderFrontMixin.front_fwd(rt, imgs.cuda(), None, 196)
    ws = rt.get_ws(rt.enc_geom, 2, 197, True)
    W, _ = rt.stack_weights("blocks.", 12)
    xe = BlockStack(rt.k, rt.enc_geom).forward(ws, x0, W)
    torch.cuda.synchronize()
    x = S._tokens(sd, imgs, cfg)
    def r(a, b): return ((a.float().cpu().reshape(b.shape) - b).abs().max() / b.abs().max()).item()
    print("x0", r(x0, x))
    for i in range(3):
        pre = f"blocks.{i}."
        bw = ws.block(i)
        ln1 = S.q(O.layer_norm(x, sd, pre + "norm1."))
        print(i, "ln1", r(bw.ln1, ln1), " frac differing", (bw.ln1.float().cpu().reshape(ln1.shape) != ln1).float().mean().item())
        qkv = S.q(S._linear(ln1, sd, pre + "attn.qkv."))
        print(i, "qkv", r(bw.qkv, qkv), " frac differing", (bw.qkv.float().cpu().reshape(qkv.shape) != qkv).float().mean().item())
        xm = x + S.attention(ln1, sd, pre + "attn.", 12)
        print(i, "x_mid", r(bw.x_mid, xm))
        x = S.block(x, sd, pre, 12)
        print(i, "x_out", r(bw.x_out, x))
    # finer: block 0 only
    x = S._tokens(sd, imgs, cfg); pre = "blocks.0."; bw = ws.block(0)
    ln1 = S.q(O.layer_norm(x, sd, pre + "norm1."))
    B, N, C = ln1.shape; H = 12; dh = 64
    qkv = S.q(S._linear(ln1, sd, pre + "attn.qkv."))
    # attention output (before proj) in sim
    qkv_ = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4); qq, kk, vv = qkv_[0], qkv_[1], qkv_[2]
    c = dh ** -0.5 * math.log2(math.e); s = qq @ kk.transpose(-2, -1)
    m = torch.full((B, H, N, 1), float("-inf")); l = torch.zeros(B, H, N, 1); o = torch.zeros(B, H, N, dh)
    for t0 in range(0, N, 32):
        st = s[..., t0:t0+32]; mn = torch.maximum(m, st.amax(-1, keepdim=True)); alpha = torch.exp2((m - mn) * c)
        p = torch.exp2((st - mn) * c); l = l * alpha + p.sum(-1, keepdim=True); o = o * alpha + S.q(p) @ vv[..., t0:t0+32, :]; m = mn
    att = S.q((o / l).transpose(1, 2).reshape(B, N, C))
    # also exact softmax attention from the same bf16 qkv (no P rounding)
    att_exact = ((s * dh ** -0.5).softmax(-1) @ vv).transpose(1, 2).reshape(B, N, C)
    # HIP attention on the SIM's qkv (isolates the attention kernel)
    qkv_dev = qkv.bfloat16().cuda().contiguous(); out_dev = torch.empty(B, N, C, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
    rt.k.attention_fwd(qkv_dev, out_dev, lse, B, N, H, dh); torch.cuda.synchronize()
    print("attn kernel vs sim (same qkv):", r(out_dev, att), " vs exact softmax:", r(out_dev, att_exact), " sim vs exact:", ((att-att_exact).abs().max()/att_exact.abs().max()).item())
    print("bw.attn vs sim att:", r(bw.attn, att))
    xm = x + S._linear(att, sd, pre + "attn.proj.")
    ln2 = S.q(O.layer_norm(xm, sd, pre + "norm2."))
    print("ln2", r(bw.ln2, ln2))
    hp = S.q(S._linear(ln2, sd, pre + "mlp.fc1."))
    print("h_pre", r(bw.h_pre, hp), " frac diff", (bw.h_pre.float().cpu().reshape(hp.shape) != hp).float().mean().item())
    g = S.q(F.gelu(hp))
    print("h_act", r(bw.h_act, g), " frac diff", (bw.h_act.float().cpu().reshape(g.shape) != g).float().mean().item())
    # HIP fc2 on the sim's g (isolates the fc2 GEMM)
    g_dev = g.bfloat16().cuda().reshape(B*N, -1).contiguous(); outf = torch.empty(B*N, C, device="cuda"); xm_dev = xm.cuda().reshape(B*N, C).contiguous()
    from ssl4polyp_amd._lib import EPI_RESIDUAL
    rt.k.linear_fwd(g_dev, W[0]["mlp.fc2.weight"], W[0]["mlp.fc2.bias"], outf, B*N, C, 3072, EPI_RESIDUAL, resid=xm_dev); torch.cuda.synchronize()
    xo = xm + S._linear(g, sd, pre + "mlp.fc2.")
    print("fc2 kernel on sim inputs vs sim:", r(outf, xo))
