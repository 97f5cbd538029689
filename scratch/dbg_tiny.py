import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssl4polyp_amd as A
from oracle import vit_mae_ref as O
fx = dict(np.load('tests/golden/tiny_mae.npz'))
cfg = O.VIT_TINY
sd = {k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('w/')}
imgs, noise = torch.from_numpy(fx['imgs']), torch.from_numpy(fx['noise'])
for prec in ('fp32',):
    m = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=1, decoder_num_heads=1, precision=prec)
    m.load_state_dict(sd); m.cuda()
    loss, pred, mask = m(imgs.cuda(), 0.75, noise=noise.cuda())
    ref = torch.from_numpy(fx['pred'])
    d = (pred.detach().cpu() - ref).abs()
    print('max err', d.max().item(), 'ref max', ref.abs().max().item())
    print('err per sample', d.amax(dim=(1,2)))
    print('err per position (sample0)', d[0].amax(dim=1))
    print('mask sample0', mask[0].cpu())
    # stage check: encoder only via oracle pieces
    ids_keep, mk, ids_restore = O.masking_from_noise(noise, 0.75)
    lat = O.mae_forward_encoder(sd, imgs, ids_keep, cfg)
    # rerun engine pieces manually
    rt = m._rt
    from ssl4polyp_amd.models import _EncoderFrontMixin
    from ssl4polyp_amd.engine import BlockStack
    cols, x0 = _EncoderFrontMixin.front_fwd(rt, imgs.cuda(), ids_keep.int().cuda().contiguous(), ids_keep.shape[1])
    # oracle x0
    x = O.patch_embed(imgs, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], 8) + sd['pos_embed'][:,1:]
    x = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1,1,64))
    x = torch.cat(((sd['cls_token']+sd['pos_embed'][:,:1]).expand(4,-1,-1), x), 1)
    print('x0 err', (x0.cpu().view(4,5,64)-x).abs().max().item(), x.abs().max().item())
    ws = rt.get_ws(rt.enc_geom, 4, 5, True)
    W,_ = rt.stack_weights('blocks.', 2)
    xe = BlockStack(rt.k, rt.enc_geom).forward(ws, x0, W)
    xo = x
    for i in range(2):
        pre=f'blocks.{i}.'
        bw = ws.block(i)
        ln1 = O.layer_norm(xo, sd, pre+'norm1.')
        print(i,'ln1 err', (bw.ln1.cpu().view(4,5,64)-ln1).abs().max().item())
        qkv = torch.nn.functional.linear(ln1, sd[pre+'attn.qkv.weight'], sd[pre+'attn.qkv.bias'])
        print(i,'qkv err', (bw.qkv.cpu().view(4,5,192)-qkv).abs().max().item(), qkv.abs().max().item())
        xm = xo + O.attention(ln1, sd, pre+'attn.', 2)
        print(i,'xmid err', (bw.x_mid.cpu().view(4,5,64)-xm).abs().max().item(), xm.abs().max().item())
        xo = O.block(xo, sd, pre, 2)
        print(i,'xout err', (bw.x_out.cpu().view(4,5,64)-xo).abs().max().item(), xo.abs().max().item())
