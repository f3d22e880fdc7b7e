"""Training entry point — the loop of reference train.py:28-81 on the MI355X build: same option surface (TrainOptions), same
model API (set_input / optimize_parameters / get_current_losses / save_networks / update_learning_rate), the reference's
console + loss_log.txt line.  Out of scope by SURVEY.md §2: visdom / HTML image pages.

    python -m nemar_amd.train --model nemar --stn_type unet --dataset_mode gpupairs --dataroot synthetic \\
        --img_height 256 --img_width 256 --crop_size 256 --load_size 286 --batch_size 8 --lambda_smooth 10
"""
import os
import sys
import time

import torch

from . import distributed as dist
from . import launch
from .data import create_dataset
from .models import create_model
from .options import TrainOptions
from .util.visualizer import LossLogger


_Options = TrainOptions


def main(argv=None):
    """`--gpu_ids 0,1,...` (reference options/base_options.py:127-135: the batch is split over the listed GPUs) = one process per
    listed GPU here: started without a torch.distributed environment the command re-executes itself once per GPU
    (nemar_amd/launch.py); under torchrun the ranks are already there.  `--batch_size` stays the GLOBAL batch, each rank takes
    batch_size / world of every batch (equal shards: mean of means == full-batch mean), gradients are averaged over RCCL inside
    optimize_parameters(), and rank 0 alone writes checkpoints and logs."""
    quiet = int(os.environ.get('RANK', '0')) != 0
    opt = _Options().parse(argv, quiet=quiet)
    ids = list(opt.gpu_ids)
    if len(ids) > 1 and not launch.under_launcher():
        raise SystemExit(launch.spawn_local_ranks(len(ids), argv=sys.argv[1:] if argv is None else list(argv), module='nemar_amd.train'))
    env_world, env_local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))
    # this rank's GPU comes from --gpu_ids when it lists one per rank; it is selected before the RCCL communicator exists
    rank, world, local = dist.init_from_env(device=ids[env_local] if (env_world > 1 and len(ids) == env_world) else None)
    if world > 1:
        if len(ids) not in (1, world):
            raise SystemExit('--gpu_ids lists %d GPUs but WORLD_SIZE=%d' % (len(ids), world))
        opt.gpu_ids = [ids[local] if len(ids) == world else local]      # this rank's GPU
        torch.cuda.set_device(opt.gpu_ids[0])
        if opt.batch_size % world:
            raise SystemExit('--batch_size %d (the global batch) is not divisible by %d ranks' % (opt.batch_size, world))
    global_batch = opt.batch_size
    opt.shard_rank, opt.shard_world = rank, world            # the loader cuts every global batch into equal per-rank shards
    dataset = create_dataset(opt)
    dataset_size = len(dataset)
    if rank == 0:
        print('The number of training images = %d' % dataset_size)
    opt.batch_size = global_batch // world                   # what the model sees per step on this rank
    # The reference is unseeded.  --seed S makes a run repeatable (weights, dropout masks, loader order); data-parallel ranks always
    # start from one seed so that they build identical replicas (setup() also broadcasts rank 0's parameters).
    seed = getattr(opt, 'seed', None)
    if seed is not None or world > 1:
        torch.manual_seed(0 if seed is None else int(seed))
    model = create_model(opt)
    model.setup(opt)
    opt.batch_size = global_batch
    logger = LossLogger(opt) if rank == 0 else None
    if rank != 0:
        model.tb_visualizer = None                           # scalar / offset reports: rank 0 only
    total_iters = 0
    if hasattr(dataset, 'epoch'):
        dataset.epoch = int(opt.epoch_count) - 1             # --continue_train resumes the shuffle sequence where it stopped
    for epoch in range(opt.epoch_count, opt.niter + opt.niter_decay + 1):
        epoch_start_time = time.time()
        iter_data_time = time.time()
        epoch_iter = 0
        t_data = 0.0
        for data in dataset:
            iter_start_time = time.time()
            if total_iters % opt.print_freq == 0:
                t_data = iter_start_time - iter_data_time
            total_iters += opt.batch_size
            epoch_iter += opt.batch_size
            model.set_input(data)
            if getattr(opt, 'step_graph', False) and world == 1 and getattr(model, '_graph', None) is None:
                model.enable_step_graph()                              # this batch and every later one: graph replays
            model.optimize_parameters()
            if total_iters % opt.print_freq == 0:
                losses = model.get_current_losses()                    # the only host synchronisation of the loop
                if world > 1:                                          # the logged losses are the global-batch means
                    t = torch.tensor([float(v) for v in losses.values()], device=model.device, dtype=torch.float64)
                    torch.distributed.all_reduce(t)
                    losses = type(losses)(zip(losses.keys(), (t / world).tolist()))
                t_comp = (time.time() - iter_start_time) / opt.batch_size
                if rank == 0:
                    logger.print_current_losses(epoch, epoch_iter, losses, t_comp, t_data)
            if total_iters % opt.save_latest_freq == 0 and rank == 0:  # replicas are identical: one writer
                print('saving the latest model (epoch %d, total_iters %d)' % (epoch, total_iters))
                model.save_networks('iter_%d' % total_iters if opt.save_by_iter else 'latest')
            iter_data_time = time.time()
        if epoch % opt.save_epoch_freq == 0 and rank == 0:
            print('saving the model at the end of epoch %d, iters %d' % (epoch, total_iters))
            model.save_networks('latest')
            model.save_networks(epoch)
        if rank == 0:
            print('End of epoch %d / %d \\t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay, time.time() - epoch_start_time))
        if model.tb_visualizer:
            model.tb_visualizer.epoch_step()
    if model.tb_visualizer:
        model.tb_visualizer.end()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
