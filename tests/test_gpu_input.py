"""Device-side input tail (pm_preprocess_u8, DevicePrefetcher) against the CPU oracle: bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("B,H,W", [(3, 224, 224), (2, 36, 20), (1, 1, 4), (5, 17, 64)])
@pytest.mark.parametrize("with_flips", [False, True])
def test_preprocess_u8_bit_exact(B, H, W, with_flips):
    from oracle.input_ref import to_tensor_normalize
    from ssl4polyp_amd.data import preprocess_u8
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    flips = torch.randint(0, 4, (B,), dtype=torch.uint8, generator=g) if with_flips else None
    want = to_tensor_normalize(x, flips)
    got = preprocess_u8(x.to(DEV), flips.to(DEV) if flips is not None else None)
    assert got.dtype == torch.float32 and got.shape == (B, 3, H, W)
    assert torch.equal(got.cpu(), want)
    # other statistics (MAE pre-train uses the same ImageNet values; check the arguments are honoured)
    mean, std = (0.5, 0.25, 0.125), (0.5, 2.0, 0.3)
    assert torch.equal(preprocess_u8(x.to(DEV), None, mean, std).cpu(), to_tensor_normalize(x, None, mean, std))


def test_preprocess_u8_rejects_bad_arguments():
    from ssl4polyp_amd._lib import PolypMaeError
    from ssl4polyp_amd.data import preprocess_u8
    with pytest.raises(ValueError):
        preprocess_u8(torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device=DEV))        # not HWC
    with pytest.raises(PolypMaeError):
        preprocess_u8(torch.zeros(1, 8, 6, 3, dtype=torch.uint8, device=DEV))        # W % 4 != 0
    with pytest.raises(PolypMaeError):
        preprocess_u8(torch.zeros(1, 8, 8, 3, dtype=torch.uint8))                    # host tensor: no CPU fallback


def test_device_prefetcher_order_values_and_flips():
    from oracle.input_ref import to_tensor_normalize
    from ssl4polyp_amd.data import DevicePrefetcher
    g = torch.Generator().manual_seed(7)
    batches = [(torch.randint(0, 256, (4, 32, 32, 3), dtype=torch.uint8, generator=g), torch.arange(4) + 10 * i)
               for i in range(5)]
    seen = 0
    for i, (imgs, labels) in enumerate(DevicePrefetcher(batches, DEV)):
        assert imgs.is_cuda and labels.is_cuda
        assert torch.equal(imgs.cpu(), to_tensor_normalize(batches[i][0])) and torch.equal(labels.cpu(), batches[i][1])
        seen += 1
    assert seen == 5
    # with flips drawn from a seeded generator: the same draws on the host reproduce the device result
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    for i, (imgs, _) in enumerate(DevicePrefetcher(batches, DEV, flip_p=0.5, generator=g1)):
        r = torch.rand(2, 4, generator=g2)
        flips = (r[0] < 0.5).to(torch.uint8) | ((r[1] < 0.5).to(torch.uint8) << 1)
        assert torch.equal(imgs.cpu(), to_tensor_normalize(batches[i][0], flips))
    # feeds the model like any loader
    import ssl4polyp_amd as A
    m = A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=1, num_heads=2, out_token="cls").to(DEV)
    big = [(torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=g), torch.zeros(2)) for _ in range(2)]
    for imgs, labels in DevicePrefetcher(big, DEV):
        assert m(imgs).shape == (2, 2)


def test_device_prefetcher_on_the_engines_side_stream_trains_the_same():
    """stream="side" (what a data-parallel rank uses: no fifth busy stream beside RCCL's): the staging thread enqueues the copies and
    the transform on the engine's weight-gradient stream while the main thread enqueues AdamW / weight gradients there -- same batches,
    same losses and same weights after a few optimizer steps as with a stream of the prefetcher's own."""
    import ssl4polyp_amd as A
    from ssl4polyp_amd import engine
    from ssl4polyp_amd.data import DevicePrefetcher
    from ssl4polyp_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randint(0, 256, (8, 224, 224, 3), dtype=torch.uint8, generator=g), torch.randint(0, 2, (8,), generator=g))
               for _ in range(6)]

    def train(mode):
        torch.manual_seed(5)
        m = A.ViT_from_MAE(None, True, 2, False, None, embed_dim=128, depth=2, num_heads=2, out_token="cls").to(DEV)
        opt = FusedAdamW(m, [{"params": list(m.parameters())}], lr=1e-3, overlap_forward=True)
        pf = DevicePrefetcher(batches, DEV, flip_p=0.5, generator=torch.Generator().manual_seed(2), stream=mode)
        losses = []
        for imgs, labels in pf:
            opt.zero_grad(set_to_none=True)
            loss = A.supervised_loss(m(imgs), labels)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        used = pf._stream
        return [float(l) for l in losses], {k: v.detach().float().cpu() for k, v in m.state_dict().items()}, used

    la, sa, st_own = train("own")
    lb, sb, st_side = train("side")
    assert st_side is engine._shared_stream(DEV, "side") and st_own is not st_side
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    with pytest.raises(ValueError):
        DevicePrefetcher(batches, DEV, stream="main")
