"""MAE pre-training entry point on the MI355X engine -- the flag surface of the reference's
src/ssl4polyp/models/mae/main_pretrain.py:48-135 (model, batch_size per GPU, epochs, accum_iter, mask_ratio,
norm_pix_loss, weight_decay, lr / blr / min_lr, warmup_epochs, output_dir, resume, seed, start_epoch) and its
behaviour: lr = blr * batch_size * accum_iter * world / 256 (:201-204), AdamW(betas=(0.9, 0.95)) with no decay on
1-D parameters (:217-218), per-iteration half-cosine schedule, checkpoint-<epoch>.pth + last.pth (misc.py:306-335).

    python -m ssl4polyp_amd.main_pretrain --synthetic 50 --epochs 2 --batch_size 256
    python -m torch.distributed.run --nproc-per-node 8 -m ssl4polyp_amd.main_pretrain --synthetic 50 ...

Data: the reference's torchvision ImageFolder + augmentation pipeline is outside this build's scope (SURVEY §8-f
rank 2); pass `--synthetic N` for N device-resident Hyperkvasir-shaped batches per epoch, or import `run()` and hand
it your own iterable of (images, _) batches.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import torch.distributed as dist

from . import models
from .optim import FusedAdamW, LossScaler, add_weight_decay
from .parallel import DataParallel
from .train import SyntheticLoader, load_mae_checkpoint, save_mae_checkpoint, train_one_epoch_mae


def get_args_parser():
    p = argparse.ArgumentParser("MAE pre-training (MI355X)", add_help=True)
    p.add_argument("--batch_size", default=64, type=int, help="batch size per GPU")
    p.add_argument("--epochs", default=400, type=int)
    p.add_argument("--accum_iter", default=1, type=int)
    p.add_argument("--model", default="mae_vit_base_patch16", type=str)
    p.add_argument("--input_size", default=224, type=int)
    p.add_argument("--mask_ratio", default=0.75, type=float)
    p.add_argument("--norm_pix_loss", action="store_true")
    p.set_defaults(norm_pix_loss=False)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--blr", type=float, default=1e-3)
    p.add_argument("--min_lr", type=float, default=0.0)
    p.add_argument("--warmup_epochs", type=int, default=40)
    p.add_argument("--output_dir", default="./output_dir")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"],
                   help="fp16 = the reference's `--precision amp` arithmetic (fp16 matmuls, f32 accumulation, dynamic loss scaling)")
    p.add_argument("--synthetic", default=0, type=int, help="number of synthetic batches per epoch")
    p.add_argument("--save_every", default=1, type=int)
    p.add_argument("--log_every", default=20, type=int)
    return p


def run(args, data_loader=None):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from .engine import reserve_streams
    reserve_streams(device)  # before RCCL creates its streams: one hardware queue per engine stream (engine.reserve_streams)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(args.seed + rank)  # main_pretrain.py:147
    model = getattr(models, args.model)(norm_pix_loss=args.norm_pix_loss, precision=args.precision)
    ddp = DataParallel(model, device)
    eff = args.batch_size * args.accum_iter * world
    if args.lr is None:
        args.lr = args.blr * eff / 256
    opt = FusedAdamW(model, add_weight_decay(model, args.weight_decay), lr=args.lr, betas=(0.9, 0.95), overlap_forward=True)
    opt.grad_sync = ddp.sync
    opt.grad_scale = 1.0 / world
    scaler = LossScaler() if args.precision == "fp16" else None  # main_pretrain.py:219: loss_scaler = NativeScaler()
    if args.resume:  # (DataParallel() above already bound the parameters to the flat device storage)
        args.start_epoch = load_mae_checkpoint(args.resume, model, opt, args, loss_scaler=scaler)
    if data_loader is None:
        if args.synthetic <= 0:
            raise SystemExit("no data: pass --synthetic N or call run(args, data_loader)")
        data_loader = SyntheticLoader(args.batch_size, args.synthetic, device, args.input_size, seed=1234 + rank, fresh=True)
    log_path = os.path.join(args.output_dir, "log.txt")
    for epoch in range(args.start_epoch, args.epochs):
        stats = train_one_epoch_mae(ddp, data_loader, opt, device, epoch, args, log_every=args.log_every, loss_scaler=scaler,
                                    printer=(lambda r: print(f"epoch {epoch} {json.dumps(r)}", flush=True)) if rank == 0 else None)
        if rank == 0:
            os.makedirs(args.output_dir, exist_ok=True)
            with open(log_path, "a") as f:  # main_pretrain.py:303-310
                f.write(json.dumps({"train_loss": stats.loss, "train_lr": stats.lr, "epoch": epoch,
                                    "samples_per_sec": stats.samples_per_sec}) + "\n")
        if (epoch + 1) % args.save_every == 0 or epoch + 1 == args.epochs:
            save_mae_checkpoint(os.path.join(args.output_dir, "ckpts"), epoch, model, opt, args,
                                scaler_state=scaler.state_dict() if scaler is not None else None)
    if world > 1:
        dist.destroy_process_group()
    return model, opt


if __name__ == "__main__":
    run(get_args_parser().parse_args())
