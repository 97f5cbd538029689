"""Does the next forward really start before the overlapped AdamW has finished?  Events around the pieces of one step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
model, ddp, opt = bench.build("cls", "bf16", dev, 1, 64)
imgs, labels = bench.make_batch("cls", 64, dev, 0)
step = bench.make_step("cls", ddp, opt, imgs, labels)
for _ in range(5): step()
torch.cuda.synchronize()
rt = model._rt
E = lambda: torch.cuda.Event(enable_timing=True)
orig_front = type(model)._classify
import ssl4polyp_amd.models as M
orig_ff = M._EncoderFrontMixin.front_fwd
marks = {}
def front_fwd(rt_, imgs_, ids, keep, pos_name="pos_embed"):
    e = E(); e.record(); marks["fwd_first_kernel_enqueued_after"] = e
    out = orig_ff(rt_, imgs_, ids, keep, pos_name)
    e2 = E(); e2.record(); marks["front_done"] = e2
    return out
M._EncoderFrontMixin.front_fwd = staticmethod(front_fwd)
for it in range(3):
    e0 = E(); e0.record()
    step()
    side = rt.k.side_stream(dev)
    e_side = E(); e_side.record(side)     # after the AdamW kernels on the side stream
    e_main = E(); e_main.record()         # main stream right after opt.step() returned
    pend = [(r, lo, hi) for r, lo, hi, ev in rt.pending_updates]
    step()
    torch.cuda.synchronize()
    print(f"iter {it}: pending updates after step(): {len(pend)} entries, first {pend[:3]} last {pend[-2:]}")
    print(f"   adamw end (side) at {e0.elapsed_time(e_side):.3f} ms, main after step() at {e0.elapsed_time(e_main):.3f} ms, "
          f"next forward's front starts at {e0.elapsed_time(marks['fwd_first_kernel_enqueued_after']):.3f} ms, front done {e0.elapsed_time(marks['front_done']):.3f} ms")
