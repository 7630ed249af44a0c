# coding=utf-8
"""Host helpers and constants with the names the reference uses (src/utils/utils.py, src/utils/global_p.py)."""
import logging
import os

import numpy as np
import torch

LOWER_METRIC_LIST = ['rmse', 'mae']

# file suffixes / dict keys (src/utils/global_p.py:19-31,49-53,84-92)
TRAIN_SUFFIX = '.train.csv'
VALIDATION_SUFFIX = '.validation.csv'
TEST_SUFFIX = '.test.csv'
INFO_SUFFIX = '.info.json'
USER_SUFFIX = '.user.csv'
ITEM_SUFFIX = '.item.csv'
TRAIN_GROUP_SUFFIX = '.train_group.csv'
VT_GROUP_SUFFIX = '.vt_group.csv'
RANK_FILE_NAME = 'rank.csv'
PROPENSITY_SUFFIX = '.propensity.npy'
EXPO_SUFFIX = '.ips_expo_prob.npy'
K_SAMPLE_ID = 'sample_id'
REAL_BATCH_SIZE = 'real_batch_size'
TOTAL_BATCH_SIZE = 'total_batch_size'


def parse_global_args(parser):
    """src/utils/utils.py:10-28 — same flags and defaults."""
    parser.add_argument('--gpu', type=str, default='0', help='Set CUDA_VISIBLE_DEVICES')
    parser.add_argument('--verbose', type=int, default=logging.INFO, help='Logging Level, 0, 10, ..., 50')
    parser.add_argument('--log_file', type=str, default='../log/log.txt', help='Logging file path')
    parser.add_argument('--result_file', type=str, default='../result/result.npy', help='Result file path')
    parser.add_argument('--random_seed', type=int, default=2019, help='Random seed of numpy and torch.')
    parser.add_argument('--train', type=int, default=1, help='To train the model or not.')
    return parser


def format_metric(metric):
    """src/utils/utils.py:64-79: floats with four decimals, ints as ints, comma separated."""
    if not isinstance(metric, (tuple, list)):
        metric = [metric]
    out = []
    for m in metric:
        if isinstance(m, (float, np.floating)):
            out.append('%.4f' % m)
        elif isinstance(m, (int, np.integer)):
            out.append('%d' % m)
    return ','.join(out)


def shuffle_in_unison_scary(data):
    """src/utils/utils.py:82-92: every array of the dict gets the same permutation (same RNG state restored)."""
    state = np.random.get_state()
    for k in data:
        np.random.set_state(state)
        np.random.shuffle(data[k])
    return data


def best_result(metric, results_list):
    """src/utils/utils.py:95-107."""
    if isinstance(metric, (list, tuple)):
        metric = metric[0]
    return min(results_list) if metric in LOWER_METRIC_LIST else max(results_list)


def strictly_increasing(l):
    return all(x < y for x, y in zip(l, l[1:]))


def strictly_decreasing(l):
    return all(x > y for x, y in zip(l, l[1:]))


def world_size():
    """Ranks of the torch.distributed job this process belongs to (1: the reference's single-GPU run)."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def is_rank0():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def dist_world_rank():
    """(world size, rank) of the process group, (1, 0) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def barrier():
    """No-op in a single-process run; with several ranks: all of them have reached this point (a file rank 0 wrote is there)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        torch.cuda.synchronize()
        dist.barrier()


class Deadline(object):
    """Host-side watchdog around a phase that may hang at first contact with a peer (the warm-up steps of a multi-rank job: the
    first collectives on real RCCL): if the phase is not over — cancel() — after `seconds`, EVERY thread's stack is irrelevant,
    the process prints what it was waiting for to stderr and exits with status 3.  Under torch.distributed.run a rank that
    exits non-zero brings the whole job down with a message instead of burning the launcher's time limit.  Never a re-exec,
    never a retry in the same process."""

    def __init__(self, seconds, what):
        import threading
        self.what, self.seconds = what, float(seconds)
        self._t = None
        if self.seconds > 0:
            self._t = threading.Timer(self.seconds, self._fire)
            self._t.daemon = True
            self._t.start()

    def _fire(self):
        import sys
        rank = os.environ.get('RANK', '0')
        sys.stderr.write('[dccf_amd] rank %s: DEADLINE of %.0f s exceeded while waiting for: %s — exiting with status 3 '
                         '(a collective that never completes usually means a peer died or the ranks disagree on the schedule; '
                         'DCCF_DIST_BACKEND=gloo rehearses the same launch without RCCL)\n' % (rank, self.seconds, self.what))
        sys.stderr.flush()
        os._exit(3)

    def cancel(self):
        if self._t is not None:
            self._t.cancel()
            self._t = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.cancel()
        return False


def free_port():
    """A TCP port nobody listens on right now, chosen by the kernel (single-process rendezvous of the one-rank pipelines)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def dist_timeout_s():
    """Timeout of every torch.distributed collective of this job (DCCF_DIST_TIMEOUT_S, default 180 s: c10d's own default is 10
    to 30 minutes, longer than a driver's whole time limit)."""
    return float(os.environ.get('DCCF_DIST_TIMEOUT_S', '180'))


def init_distributed():
    """One process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment): binds this
    process to its GPU and creates the process group — backend "nccl" = RCCL over xGMI; DCCF_DIST_BACKEND=gloo rehearses on a
    box with fewer GPUs than ranks (the ranks then share the visible devices).  Returns (rank, world size).  Fails early and
    loudly: fewer visible GPUs than local ranks is an error here, not a hang inside the first collective, and every collective
    of the group times out after dist_timeout_s()."""
    import datetime
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    backend = os.environ.get('DCCF_DIST_BACKEND', 'nccl')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and ndev < local_world:
        raise RuntimeError('%d ranks on this node but only %d visible GPUs: the RCCL backend needs one GPU per rank '
                           '(LOCAL_WORLD_SIZE=%d, LOCAL_RANK=%d; set DCCF_DIST_BACKEND=gloo to rehearse with shared devices)'
                           % (local_world, ndev, local_world, local))
    if backend != 'nccl':
        local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=dist_timeout_s()), **kw)
    return dist.get_rank(), dist.get_world_size()


def device():
    if not torch.cuda.is_available():
        raise RuntimeError('dccf_amd needs an MI355X (no CPU fallback): torch.cuda.is_available() is False')
    return torch.device('cuda', torch.cuda.current_device())


def numpy_to_torch(d):
    """src/utils/utils.py:154-163: numpy -> tensor in HBM."""
    return torch.from_numpy(np.ascontiguousarray(d)).to(device())


def check_dir_and_mkdir(path):
    """src/utils/utils.py:170-178."""
    if os.path.basename(path).find('.') == -1 or path.endswith('/'):
        dirname = path
    else:
        dirname = os.path.dirname(path)
    if dirname and not os.path.exists(dirname):
        os.makedirs(dirname, exist_ok=True)
