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


def test_training_monitor_reports_weight_histograms(tmp_path):
    """reference util/tb_visualizer.py:34-40,79-80: '<net>/data/Weight|Bias/<name>' histograms of every trainable parameter with each
    report, unless --tbvis_disable_report_weights."""
    import argparse
    import json
    import torch
    from nemar_amd.util import visualizer as V

    class Model:
        device = torch.device('cpu')
        netT = torch.nn.Conv2d(2, 3, 1)
        netR = torch.nn.Linear(4, 2)
        netD = torch.nn.Conv2d(1, 1, 1, bias=False)

        def get_current_losses(self):
            return {'L1_TR': 1.5}

    for p in Model.netD.parameters():
        p.requires_grad_(False)
    for disable in (False, True):
        opt = argparse.Namespace(checkpoints_dir=str(tmp_path), name='h%d' % disable, tbvis_iteration_update_rate=2,
                                 tbvis_disable_report_offsets=True, tbvis_disable_report_weights=disable)
        mon = V.TrainingMonitor(Model(), opt)
        for _ in range(3):
            mon.iteration_step()
        mon.end()
        path = os.path.join(mon.log.dir, 'histograms.jsonl')
        if disable:
            assert not os.path.exists(path)
            continue
        rows = [json.loads(l) for l in open(path)]
        tags = [r['tag'] for r in rows if r['step'] == 0]
        assert tags == ['netR/data/Weight/weight', 'netR/data/Bias/bias', 'netT/data/Weight/weight', 'netT/data/Bias/bias']   # frozen D: none
        assert {r['step'] for r in rows} == {0, 1}            # iterations 0 and 2
        w = Model.netR.weight.detach()
        r0 = rows[0]
        assert sum(r0['counts']) == w.numel() and abs(r0['min'] - float(w.min())) < 1e-6 and abs(r0['max'] - float(w.max())) < 1e-6
