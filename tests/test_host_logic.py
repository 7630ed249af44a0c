# coding=utf-8
"""CPU: the host side of the product (data loader / processor / metrics / CLI surface) against the reference's golden
vectors.  Integer paths bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, REPO
from dccf_amd import utils, rank_metrics
from dccf_amd.data_loader import DataLoader
from dccf_amd.data_processor import DataProcessor
from dccf_amd.models import RecModel


@pytest.fixture
def cpu_tensors(monkeypatch):
    monkeypatch.setattr(utils, 'device', lambda: torch.device('cpu'))


def write_split(tmp_path, g):
    d = tmp_path / 'toyds'
    d.mkdir()
    for nm in ('train', 'validation', 'test'):
        np.savetxt(str(d / ('toyds.%s.csv' % nm)), g['df/' + nm], fmt='%d', delimiter=',')
    return str(tmp_path)


def test_loader_and_processor_reproduce_reference_batches(tmp_path, cpu_tensors):
    g = load_golden('batches')
    path = write_split(tmp_path, g)
    dl = DataLoader(path=path, dataset='toyds', label='label', sep=',')
    assert dl.user_num == int(g['user_num']) and dl.item_num == int(g['item_num'])
    assert os.path.exists(dl.info_file) and os.path.exists(dl.train_his_file) and os.path.exists(dl.vt_his_file)
    dl2 = DataLoader(path=path, dataset='toyds', label='label', sep=',')        # second load reads the cached files
    assert dl2.train_user_his == dl.train_user_his and dl2.column_max == dl.column_max
    dl.drop_neg()
    dp = DataProcessor(dl, RecModel, rank=1, test_neg_n=int(g['test_neg_n']))
    np.random.seed(int(g['np_seed']))
    test = dp.get_test_data()
    train = dp.get_train_data(epoch=-1)
    val = dp.get_validation_data()
    for nm, d in (('test', test), ('validation', val)):
        for k in ('uid', 'iid', 'Y', 'X', 'sample_id'):
            assert np.array_equal(np.asarray(d[k]), g['%s/%s' % (nm, k)]), (nm, k)
    for ep in range(2):
        tr = dp.get_train_data(epoch=ep)
        for k in ('uid', 'iid', 'Y', 'X', 'sample_id'):
            assert np.array_equal(np.asarray(tr[k]), g['train_ep%d/%s' % (ep, k)]), (ep, k)
        batches = dp.prepare_batches(tr, int(g['batch_size']), train=True)
        assert len(batches) == int(g['train_ep%d/n_batches' % ep])
        assert np.array_equal(np.concatenate([b['X'].numpy() for b in batches]), g['train_ep%d/batch_X' % ep])
        assert np.array_equal(np.concatenate([b['Y'].numpy() for b in batches]), g['train_ep%d/batch_Y' % ep])
        assert np.array_equal(np.concatenate([b['sample_id'] for b in batches]), g['train_ep%d/batch_sample_id' % ep])
        assert np.array_equal(np.array([b['real_batch_size'] for b in batches]), g['train_ep%d/batch_sizes' % ep])
    vb = dp.prepare_batches(val, 64, train=False)
    assert np.array_equal(np.concatenate([b['X'].numpy() for b in vb]), g['validation/batch_X'])
    assert dp.prepare_batches(val, 64, train=False) is vb      # cached like the reference's vt_batches_buffer


def test_metrics_match_reference():
    g = load_golden('metrics')
    vals = rank_metrics.evaluate_method(g['p'], {'uid': g['uid'], 'Y': g['Y']}, [str(m) for m in g['metrics']])
    assert np.allclose(vals, g['values'], rtol=1e-6, atol=1e-9)
    r = [3, 2, 3, 0, 0, 1, 2, 2, 3, 0]
    assert abs(rank_metrics.dcg_at_k(r, 2, method=1) - 4.2618595071429155) < 1e-12
    assert abs(rank_metrics.ndcg_at_k([2, 1, 2, 0], 4, method=1) - 0.96519546960144276) < 1e-12
    assert rank_metrics.precision_at_k([0, 0, 1], 3) == pytest.approx(1 / 3)
    with pytest.raises(ValueError):
        rank_metrics.precision_at_k([0, 0, 1], 4)
    # users with fewer candidates than k, and a user without positives (ndcg 0)
    p = np.array([0.9, 0.1, 0.5, 0.4, 0.3])
    uid = np.array([1, 1, 2, 2, 2])
    y = np.array([0., 1., 0., 0., 0.])
    assert rank_metrics.evaluate_method(p, {'uid': uid, 'Y': y}, ['ndcg@5'])[0] == pytest.approx(0.5 * (1 / np.log2(3)))


def test_cli_flags_match_reference_defaults():
    """Same flag names and defaults as the classes of the reference contribute."""
    import argparse
    from dccf_amd.models import DCCF, IPSBiasedMF
    from dccf_amd.runner import BaseRunner
    p = argparse.ArgumentParser()
    utils.parse_global_args(p)
    DataLoader.parse_data_args(p)
    DCCF.parse_model_args(p, 'DCCF')
    BaseRunner.parse_runner_args(p)
    DataProcessor.parse_dp_args(p)
    a = p.parse_args([])
    assert (a.random_seed, a.gpu, a.train) == (2019, '0', 1)
    assert (a.path, a.sep, a.label) == ('../datasets/', ',', 'label')
    assert (a.u_vector_size, a.i_vector_size, a.n_layers) == (64, 64, 1)
    assert (a.sentence_model, a.sample_num, a.attribute_num, a.std) == ('paraphrase-distilroberta-base-v1', 10, 2, 0.1)
    assert (a.epoch, a.lr, a.batch_size, a.eval_batch_size, a.dropout, a.l2, a.optimizer, a.metric) == \
           (100, 0.01, 128, 16384, 0.2, 1e-4, 'GD', 'RMSE')
    assert (a.test_neg_n, a.load, a.check_epoch, a.early_stop, a.skip_eval) == (100, 0, 1, 1, 0)
    assert a.model_path == '../model/DCCF/DCCF.pt'
    p2 = argparse.ArgumentParser()
    IPSBiasedMF.parse_model_args(p2, 'IPSBiasedMF')
    assert p2.parse_args([]).M == 0.1


def test_product_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        RecModel(label_min=0, label_max=1, feature_num=0, user_num=3, item_num=3, u_vector_size=4, i_vector_size=4,
                 random_seed=1, model_path='/tmp/x.pt')


def test_deadline_exits_nonzero_with_a_message_and_can_be_cancelled():
    """utils.Deadline (the host-side watchdog around the warm-up of a multi-rank job, first contact with real RCCL): when the
    phase does not end in time the process says what it waited for on stderr and exits with status 3 — never a hang, never a
    re-exec; a cancelled deadline does nothing."""
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); from dccf_amd import utils\n"
            "d = utils.Deadline(float(sys.argv[1]), 'the warm-up of a test')\n"
            "if sys.argv[2] == 'cancel': d.cancel()\n"
            "time.sleep(float(sys.argv[3])); print('survived')" % REPO)
    r = subprocess.run([sys.executable, '-c', code, '0.3', 'keep', '5'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 3 and b'DEADLINE' in r.stderr and b'the warm-up of a test' in r.stderr and b'survived' not in r.stdout
    r = subprocess.run([sys.executable, '-c', code, '0.3', 'cancel', '0.8'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0 and b'survived' in r.stdout


def test_init_distributed_refuses_fewer_gpus_than_local_ranks(monkeypatch):
    """The RCCL backend needs one GPU per local rank: an early, readable error instead of a hang inside the first collective."""
    import pytest
    import torch
    from dccf_amd import utils
    monkeypatch.setenv('WORLD_SIZE', '64')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '64')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.delenv('DCCF_DIST_BACKEND', raising=False)
    assert torch.cuda.device_count() < 64
    with pytest.raises(RuntimeError, match='one GPU per rank'):
        utils.init_distributed()


def test_dist_world_rank_without_a_process_group():
    """utils.dist_world_rank (what the sharded evaluation of runner.predict_device deals its batches by): (1, 0) in a
    single-process run — the reference's evaluate (src/runners/BaseRunner.py:134-157) is single-process."""
    from dccf_amd import utils
    assert utils.dist_world_rank() == (1, 0)
    assert utils.is_rank0()
