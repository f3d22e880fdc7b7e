"""Console / file loss log and scalar log — the observability surface of the reference that a training loop needs
(SURVEY.md §8 f4), without per-iteration host synchronisation:

  * LossLogger.print_current_losses reproduces the reference's line byte for byte (util/visualizer.py:211-227):
    '(epoch: %d, iters: %d, time: %.3f, data: %.3f) ' + '%s: %.3f ' per loss, printed and appended to loss_log.txt;
  * ScalarLog + OffsetMeter reproduce what TensorboardVisualizer reports (util/tb_visualizer.py:53-92): 'loss/<name>',
    'offset/mean_x', 'offset/mean_y' every --tbvis_iteration_update_rate iterations.  The reference copies the whole
    deformation field to the host every iteration (:71-74); here the per-channel sums are accumulated ON THE DEVICE by
    the bias-gradient reduction kernel into two floats and read back only when a report is written.
visdom / HTML image pages stay out of scope."""
import json
import os
import time

import torch

from .. import ops


class LossLogger:
    def __init__(self, opt):
        self.log_name = os.path.join(opt.checkpoints_dir, opt.name, 'loss_log.txt')
        os.makedirs(os.path.dirname(self.log_name), exist_ok=True)
        with open(self.log_name, "a") as f:
            f.write('================ Training Loss (%s) ================\n' % time.strftime("%c"))

    def print_current_losses(self, epoch, iters, losses, t_comp, t_data):
        message = '(epoch: %d, iters: %d, time: %.3f, data: %.3f) ' % (epoch, iters, t_comp, t_data)
        for k, v in losses.items():
            message += '%s: %.3f ' % (k, v)
        print(message)
        with open(self.log_name, "a") as f:
            f.write('%s\n' % message)
        return message


class ScalarLog:
    """add_scalar(tag, value, step): torch.utils.tensorboard.SummaryWriter when it is importable, and always a JSON-lines
    file next to the checkpoints (the build image has no tensorboard)."""

    def __init__(self, opt):
        self.dir = '{}/{}/{}_tensorboard_logs'.format(opt.checkpoints_dir, opt.name, opt.name)
        os.makedirs(self.dir, exist_ok=True)
        self.path = os.path.join(self.dir, 'scalars.jsonl')
        self.writer = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(self.dir)
        except Exception:
            pass

    def add_scalar(self, tag, value, step):
        with open(self.path, 'a') as f:
            f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')
        if self.writer is not None:
            self.writer.add_scalar(tag, float(value), step)

    def add_histogram(self, tag, tensor, step, bins=30):
        """reference tb_visualizer.py:39 `writer.add_histogram(name, p.clone().cpu().data.numpy(), step)`: the SummaryWriter call
        when tensorboard is importable; the JSON-lines file always gets a 30-bin histogram computed ON the device (one
        torch.histc + min / max / mean: 33 floats cross PCIe instead of the whole parameter)."""
        t = tensor.detach().float().reshape(-1)
        lo, hi = float(t.min()), float(t.max())
        counts = torch.histc(t, bins=bins, min=lo, max=hi if hi > lo else lo + 1.0).tolist()
        with open(os.path.join(self.dir, 'histograms.jsonl'), 'a') as f:
            f.write(json.dumps({'tag': tag, 'step': int(step), 'min': lo, 'max': hi, 'mean': float(t.mean()), 'counts': counts}) + '\n')
        if self.writer is not None:
            self.writer.add_histogram(tag, t.cpu().numpy(), step)


class OffsetMeter:
    """Running mean of the deformation offsets per direction (reference tb_visualizer.py:59-74): update() adds the field's
    per-channel sums into a 2-float device accumulator (one kernel pair, no sync); means() reads it back and resets."""

    def __init__(self, device):
        self.acc = torch.zeros(2, dtype=torch.float32, device=device)
        self.weight = 0.0          # sum over updates of 1 (the reference averages per-iteration means)
        self.norm = []

    def update(self, offsets):
        n, c, h, w = offsets.shape
        assert c == 2
        # acc[c] += sum_{n,h,w} offsets / (n*h*w)  ==  += mean(offsets[:, c]) : scale folded in by accumulating raw sums of a
        # constant-size field; sizes may vary between calls, so keep the per-call normaliser on the host
        part = torch.zeros(2, dtype=torch.float32, device=offsets.device)
        ops._bias_grad(offsets.contiguous(), part, n, 2, h * w, ops._stream())
        self.acc.add_(part, alpha=1.0 / (n * h * w))
        self.weight += 1.0

    def means(self):
        if self.weight == 0:
            return 0.0, 0.0
        mx, my = (self.acc / self.weight).tolist()           # the only device->host transfer
        self.acc.zero_()
        self.weight = 0.0
        return mx, my


class TrainingMonitor:
    """What train.py drives every iteration (reference TensorboardVisualizer.iteration_step, tb_visualizer.py:68-85)."""

    def __init__(self, model, opt):
        self.model, self.opt = model, opt
        self.rate = int(getattr(opt, 'tbvis_iteration_update_rate', 1000))
        self.report_offsets = not getattr(opt, 'tbvis_disable_report_offsets', False)
        self.report_weights = not getattr(opt, 'tbvis_disable_report_weights', False)
        self.log = ScalarLog(opt)
        self.meter = OffsetMeter(model.device)
        self.iteration_cnt = 0
        self.save_count = 0

    def iteration_step(self):
        field = getattr(self.model, 'deformation_field_A_to_B', None)
        if self.report_offsets and field is not None and field.dim() == 4 and field.size(1) == 2:
            self.meter.update(field)
        if self.rate <= 0:
            return
        if self.iteration_cnt == 0:
            for name, v in self.model.get_current_losses().items():
                self.log.add_scalar('loss/{}'.format(name), v, self.save_count)
            if self.report_weights:
                self.save_current_weights()
            if self.report_offsets:
                mx, my = self.meter.means()
                self.log.add_scalar('offset/mean_x', mx, self.save_count)
                self.log.add_scalar('offset/mean_y', my, self.save_count)
            self.save_count += 1
        self.iteration_cnt = (self.iteration_cnt + 1) % self.rate

    def save_current_weights(self):
        """'<net>/data/Weight|Bias/<parameter name>' histograms of every trainable parameter of T, R, D (reference
        tb_visualizer.py:34-40; gated by --tbvis_disable_report_weights like there)."""
        for net_name in ('netR', 'netT', 'netD'):             # the reference's list and order (models/nemar_model.py:58)
            net = getattr(self.model, net_name, None)
            if net is None:
                continue
            for n, p in net.named_parameters():
                if p.requires_grad:
                    self.log.add_histogram('{}/data/{}/{}'.format(net_name, 'Bias' if 'bias' in n else 'Weight', n), p, self.save_count)

    def epoch_step(self):
        if self.rate > 0:          # iteration-resolution reporting is on: nothing per epoch (reference :87-95)
            return
        for name, v in self.model.get_current_losses().items():
            self.log.add_scalar('loss/{}'.format(name), v, self.save_count)
        if self.report_weights:
            self.save_current_weights()
        if self.report_offsets:
            mx, my = self.meter.means()
            self.log.add_scalar('offset/mean_x', mx, self.save_count)
            self.log.add_scalar('offset/mean_y', my, self.save_count)
        self.save_count += 1

    def end(self):
        if self.log.writer is not None:
            self.log.writer.close()
