"""Training entry point — the loop of reference train.py:28-81 on the MI355X build: same option surface (TrainOptions), same
model API (set_input / optimize_parameters / get_current_losses / save_networks / update_learning_rate), the reference's
console + loss_log.txt line.  Out of scope by SURVEY.md §2: visdom / HTML image pages.

    python -m nemar_amd.train --model nemar --stn_type unet --dataset_mode gpupairs --dataroot synthetic \\
        --img_height 256 --img_width 256 --crop_size 256 --load_size 286 --batch_size 8 --lambda_smooth 10
"""
import time

from .data import create_dataset
from .models import create_model
from .options import TrainOptions
from .util.visualizer import LossLogger


_Options = TrainOptions


def main(argv=None):
    opt = _Options().parse(argv)
    dataset = create_dataset(opt)
    dataset_size = len(dataset)
    print('The number of training images = %d' % dataset_size)
    model = create_model(opt)
    model.setup(opt)
    logger = LossLogger(opt)
    total_iters = 0
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
            model.optimize_parameters()
            if total_iters % opt.print_freq == 0:
                losses = model.get_current_losses()                    # the only host synchronisation of the loop
                t_comp = (time.time() - iter_start_time) / opt.batch_size
                logger.print_current_losses(epoch, epoch_iter, losses, t_comp, t_data)
            if total_iters % opt.save_latest_freq == 0:
                print('saving the latest model (epoch %d, total_iters %d)' % (epoch, total_iters))
                model.save_networks('iter_%d' % total_iters if opt.save_by_iter else 'latest')
            iter_data_time = time.time()
        if epoch % opt.save_epoch_freq == 0:
            print('saving the model at the end of epoch %d, iters %d' % (epoch, total_iters))
            model.save_networks('latest')
            model.save_networks(epoch)
        print('End of epoch %d / %d \\t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay, time.time() - epoch_start_time))
        if model.tb_visualizer:
            model.tb_visualizer.epoch_step()
    if model.tb_visualizer:
        model.tb_visualizer.end()


if __name__ == '__main__':
    main()
