"""CPU tier for the rows SURVEY.md §8 marks 'next' f3 / f4: the loss-log line is the reference's, character for character; the
dataset plug-in discovery follows the reference's naming rule."""
import argparse
import os

import pytest


def test_loss_log_line_is_the_reference_format(tmp_path, capsys):
    """reference util/visualizer.py:220-227:  '(epoch: %d, iters: %d, time: %.3f, data: %.3f) ' + '%s: %.3f ' per loss."""
    from collections import OrderedDict
    from nemar_amd.util.visualizer import LossLogger
    opt = argparse.Namespace(checkpoints_dir=str(tmp_path), name='exp')
    lg = LossLogger(opt)
    losses = OrderedDict([('L1_TR', 61.426683), ('GAN_TR', 0.76853), ('D', 1.13119)])
    msg = lg.print_current_losses(3, 128, losses, 0.00874, 0.0001234)
    want = '(epoch: 3, iters: 128, time: 0.009, data: 0.000) L1_TR: 61.427 GAN_TR: 0.769 D: 1.131 '
    assert msg == want
    assert capsys.readouterr().out.strip() == want.strip()
    lines = open(os.path.join(str(tmp_path), 'exp', 'loss_log.txt')).read().splitlines()
    assert lines[0].startswith('================ Training Loss (') and lines[1] == want


def test_dataset_discovery_rule():
    from nemar_amd.data import find_dataset_using_name, BaseDataset
    cls = find_dataset_using_name('gpupairs')
    assert cls.__name__ == 'GpuPairsDataset' and issubclass(cls, BaseDataset)
    with pytest.raises(ModuleNotFoundError):
        find_dataset_using_name('no_such')
