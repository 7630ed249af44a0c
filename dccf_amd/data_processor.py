# coding=utf-8
"""DataProcessor with the reference's surface (src/data_processor/DataProcessor.py:15-524).

Two training feeds:
  * ``prepare_batches(data, batch_size, train=True)`` — the reference's host path, restated: one numpy-RNG negative
    per train row with the per-epoch ``tmp_history`` rule (:446-524), feed dicts ``X=[pos;neg]`` (:160-207).  For the
    same numpy seed and call order it yields the reference's batches bit for bit (tests/test_host_logic.py).
  * ``device_epoch(epoch)`` — the MI355X path: the train set, its per-user row lists and sorted histories live in HBM
    (CSR), negatives for the whole epoch come from ONE kernel (``dccf_sample_train_negatives``: Philox streams keyed
    by (seed, epoch, uid)), and the epoch's batches are views of device tensors — no per-epoch host loop, no H2D.
Evaluation negatives (:73-111, :408-444) are sampled once per run on the host exactly as the reference does.
"""
import logging
import os
from collections import defaultdict

import numpy as np
import torch

from dccf_amd import utils, _lib


class DataProcessor(object):
    data_columns = ['X']

    @staticmethod
    def parse_dp_args(parser):
        parser.add_argument('--test_neg_n', type=int, default=100,
                            help='Negative sample num for each instance in test/validation set.')
        return parser

    def __init__(self, data_loader, model, rank, test_neg_n, seed=2019, fused_eval=False):
        """fused_eval: draw the evaluation negatives on the GPU (dccf_sample_eval_negatives) instead of the reference's
        per-user Python loop (:446-524) — minutes of host time per split at Electronics size."""
        self.data_loader, self.model, self.rank, self.test_neg_n = data_loader, model, rank, test_neg_n
        self.fused_eval = fused_eval
        self.train_data, self.validation_data, self.test_data = None, None, None
        self.seed = seed
        if self.rank == 1:
            self.train_history_dict = defaultdict(set)
            for uid, items in data_loader.train_user_his.items():
                self.train_history_dict[uid] = set(items)
            self.vt_history_dict = defaultdict(set)
            for uid, items in data_loader.vt_user_his.items():
                self.vt_history_dict[uid] = set(items)
        self.vt_batches_buffer = {}
        self._dev_eval = {}
        self._dev = None

    # ------------------------------------------------------------------ data dicts
    def format_data_dict(self, df):
        """src/data_processor/DataProcessor.py:292-356 for the id-only models of this path (append_id, no side
        features): X = [uid, iid] int64, Y = label fp32."""
        model, dl = self.model, self.data_loader
        if (dl.user_df is not None and model.include_user_features) or (dl.item_df is not None and model.include_item_features) \
                or model.include_context_features or model.include_id or not model.append_id:
            raise NotImplementedError('side-feature models are outside the DCCF hot path')
        data = {'uid': df['uid'].values, 'iid': df['iid'].values}
        if dl.label in df.columns:
            data['Y'] = np.array(df[dl.label], dtype=np.float32)
        else:
            logging.warning('No Labels In Data: ' + dl.label)
            data['Y'] = np.zeros(len(df), dtype=np.float32)
        data['X'] = df[['uid', 'iid']].values.astype(int)
        assert len(data['X']) == len(data['Y'])
        return data

    def get_train_data(self, epoch):
        """:56-70 — shuffled in unison for epoch >= 0."""
        if self.train_data is None or epoch < 0:
            logging.info('Prepare Train Data...')
            self.train_data = self.format_data_dict(self.data_loader.train_df)
            self.train_data[utils.K_SAMPLE_ID] = np.arange(0, len(self.train_data['Y']))
        if epoch >= 0:
            utils.shuffle_in_unison_scary(self.train_data)
        return self.train_data

    def _eval_data_device(self, df, tag):
        """_eval_data with the negatives drawn on the GPU: same first-occurrence rule (:420-426), same admissible set
        (outside train + validation/test history, distinct), Philox stream `tag` instead of numpy's global generator."""
        dl = self.data_loader
        uid, iid = df['uid'].values.astype(np.int64), df['iid'].values.astype(np.int64)
        _, first = np.unique(uid, return_index=True)
        users = uid[np.sort(first)]                              # distinct users in first-occurrence order
        if getattr(self, '_eval_hist', None) is None:            # CSR of train + validation/test history, items sorted
            hu, hi = [], []
            for d in (self.train_history_dict, self.vt_history_dict):
                for u, items in d.items():
                    hu.append(np.full(len(items), u, dtype=np.int64))
                    hi.append(np.fromiter(items, dtype=np.int64, count=len(items)))
            hu = np.concatenate(hu) if hu else np.zeros(0, np.int64)
            hi = np.concatenate(hi) if hi else np.zeros(0, np.int64)
            key = np.unique(hu * dl.item_num + hi)
            indptr = np.searchsorted(key // dl.item_num, np.arange(dl.user_num + 1)).astype(np.int64)
            dev = utils.device()
            self._eval_hist = (torch.as_tensor(indptr).to(dev), torch.as_tensor((key % dl.item_num).astype(np.int64)).to(dev))
        indptr, items = self._eval_hist
        negs = _lib.sample_eval_negatives(torch.as_tensor(users).to(indptr.device), indptr, items, dl.item_num,
                                          self.test_neg_n, self.seed, tag).cpu().numpy()
        assert negs.min() >= 0, 'a user has fewer than test_neg_n admissible items'          # :488
        label = np.array(df[dl.label], dtype=np.float32) if dl.label in df.columns else np.zeros(len(df), np.float32)
        n_uid, n_iid = np.repeat(users, self.test_neg_n), negs.reshape(-1)
        data = {'uid': np.concatenate([uid, n_uid]), 'iid': np.concatenate([iid, n_iid]),
                'Y': np.concatenate([label, np.zeros(len(n_uid), np.float32)])}
        data['X'] = np.stack([data['uid'], data['iid']], 1)
        data[utils.K_SAMPLE_ID] = np.arange(0, len(data['Y']))
        return data

    def _eval_data(self, df, tag=1):
        import pandas as pd
        if self.rank == 1 and self.fused_eval:
            return self._eval_data_device(df, tag)
        if self.rank == 1:
            neg_df = self.generate_neg_df(df['uid'].tolist(), df['iid'].tolist(), df, self.test_neg_n, train=False)
            df = pd.concat([df, neg_df], ignore_index=True)
        data = self.format_data_dict(df)
        data[utils.K_SAMPLE_ID] = np.arange(0, len(data['Y']))
        return data

    def get_validation_data(self):
        if self.validation_data is None:
            logging.info('Prepare Validation Data...')
            self.validation_data = self._eval_data(self.data_loader.validation_df, tag=1)
        return self.validation_data

    def get_test_data(self):
        if self.test_data is None:
            logging.info('Prepare Test Data...')
            self.test_data = self._eval_data(self.data_loader.test_df, tag=2)
        return self.test_data

    # ------------------------------------------------------------------ negatives (host, reference algorithm)
    def generate_neg_df(self, uid_list, iid_list, df, neg_n, train):
        """:408-444.  Eval: one set of neg_n negatives per DISTINCT user (first occurrence, :420-426)."""
        import pandas as pd
        if not train:
            seen, fu, fi = set(), [], []
            for u, i in zip(uid_list, iid_list):
                if u not in seen:
                    seen.add(u)
                    fu.append(u)
                    fi.append(i)
        else:
            fu, fi = uid_list, iid_list
        uids, negs, poss = self._sample_neg_from_uid_list(fu, fi, neg_n, train)
        neg_df = pd.DataFrame({'uid': uids, 'iid_neg': negs, 'iid': poss})
        neg_df = pd.merge(neg_df, df, on=['uid', 'iid'], how='left')
        neg_df = neg_df.drop_duplicates(subset=['uid', 'iid_neg', 'iid']).drop(columns=['iid'])
        neg_df = neg_df.rename(columns={'iid_neg': 'iid'})[df.columns]
        neg_df[self.data_loader.label] = 0
        return neg_df

    def _sample_neg_from_uid_list(self, uids, iids, neg_n, train):
        """:446-524 — same draws from the global numpy RNG, same rejection sets, same low-remaining fallback."""
        item_num = self.data_loader.item_num
        u_out, n_out, p_out = [], [], []
        tmp = defaultdict(set)
        for idx, uid in enumerate(uids):
            if train:
                inter = self.train_history_dict[uid] | tmp[uid]
            else:
                inter = self.train_history_dict[uid] | self.vt_history_dict[uid] | tmp[uid]
            remain_n = item_num - len(inter)
            assert remain_n >= neg_n
            if 1.0 * remain_n / item_num < 0.2:
                remain = [i for i in range(1, item_num) if i not in inter]
                picks = np.random.choice(remain, neg_n, replace=False)
                n_out.extend(picks)
                tmp[uid].update(picks)
            else:
                mine = tmp[uid]
                for _ in range(neg_n):
                    iid = np.random.randint(item_num)
                    while iid in inter or iid in mine:
                        iid = np.random.randint(item_num)
                    n_out.append(iid)
                    mine.add(iid)
            u_out.extend([uid] * neg_n)
            p_out.extend([iids[idx]] * neg_n)
            if not train:
                tmp = defaultdict(set)
        return u_out, n_out, p_out

    # ------------------------------------------------------------------ host batches (reference layout)
    def _feed_rt(self, data, b0, batch_size, train):
        b1 = min(len(data['X']), b0 + batch_size)
        fd = {'train': train, 'rank': 0, utils.K_SAMPLE_ID: data[utils.K_SAMPLE_ID][b0:b1],
              'Y': utils.numpy_to_torch(data['Y'][b0:b1]) if 'Y' in data else utils.numpy_to_torch(np.zeros(b1 - b0, np.float32)),
              'X': utils.numpy_to_torch(data['X'][b0:b1])}
        return fd

    def _feed_rk(self, data, b0, batch_size, train, neg_data):
        """:160-207."""
        if not train:
            fd = self._feed_rt(data, b0, batch_size, train)
            fd['rank'] = 1
            return fd
        b1 = min(len(data['X']), b0 + batch_size)
        real = b1 - b0
        y = np.concatenate([np.ones(real, dtype=np.float32), np.zeros(real, dtype=np.float32)])
        sid = data[utils.K_SAMPLE_ID][b0:b1]
        return {'train': True, 'rank': 1, 'Y': utils.numpy_to_torch(y),
                utils.K_SAMPLE_ID: np.concatenate([sid, sid + len(self.train_data['Y'])]),
                utils.REAL_BATCH_SIZE: real, utils.TOTAL_BATCH_SIZE: real * 2,
                'X': utils.numpy_to_torch(np.concatenate([data['X'][b0:b1], neg_data['X'][b0:b1]]))}

    def prepare_batches(self, data, batch_size, train):
        """:252-275 (+ :209-250).  Validation/test batch lists are cached as in the reference."""
        if data is None:
            return None
        key = ''
        if data is self.validation_data:
            key = 'validation_' + str(batch_size)
        elif data is self.test_data:
            key = 'test_' + str(batch_size)
        if key in self.vt_batches_buffer:
            return self.vt_batches_buffer[key]
        n = len(data['X'])
        assert n > 0
        neg_data = None
        if self.rank == 1 and train:
            neg_df = self.generate_neg_df(data['uid'], data['iid'], self.data_loader.train_df, 1, train=True)
            neg_data = self.format_data_dict(neg_df)
        batches = []
        for b0 in range(0, n, batch_size):
            if self.rank == 1:
                batches.append(self._feed_rk(data, b0, batch_size, train, neg_data))
            else:
                batches.append(self._feed_rt(data, b0, batch_size, train))
        if key:
            self.vt_batches_buffer[key] = batches
        return batches

    # ------------------------------------------------------------------ device-resident eval split
    def device_eval_set(self, data):
        """The eval split as device tensors + the per-user CSR the ranking kernel walks (DeviceEvalSet).  Validation and
        test sets are fixed for a run (:73-111) and cached; the train dict is reshuffled in place every epoch, so it is
        rebuilt (only rmse/mae are evaluated on it, BaseRunner.py:226,256 — no CSR needed)."""
        key = 'validation' if data is self.validation_data else 'test' if data is self.test_data else None
        if key is not None and key in self._dev_eval:
            return self._dev_eval[key]
        es = DeviceEvalSet(data, with_groups=key is not None or self.rank == 1)
        if key is not None:
            self._dev_eval[key] = es
        return es

    # ------------------------------------------------------------------ device-resident epoch (fused negatives)
    def device_epoch(self, epoch, batch_size):
        """The epoch's batches as device tensors: (full [nb, 2B, 2] int64, tail [2r, 2] or None).  See DeviceTrainSet."""
        if self._dev is None:
            dl = self.data_loader
            tr = dl.train_df
            pos = tr[tr[dl.label] > 0] if dl.label in tr.columns else tr
            self._dev = DeviceTrainSet(tr['uid'].values, tr['iid'].values, dl.user_num, dl.item_num, self.seed,
                                       hist_uid=pos['uid'].values, hist_iid=pos['iid'].values)
        return self._dev.epoch_batches(epoch, batch_size)


class DeviceEvalSet(object):
    """One eval split resident in HBM: X [n, 2] int64, Y [n] fp32 in sample-id order, and the CSR (indptr, rows) of each
    user's rows, in row order, that rank_eval_topk walks instead of DataFrame.sort_values + groupby('uid')
    (src/models/BaseModel.py:83-88)."""

    def __init__(self, data, with_groups=True):
        dev = utils.device()
        self.n = len(data['Y'])
        self.X = torch.as_tensor(np.ascontiguousarray(data['X']), dtype=torch.int64).to(dev)
        self.Y = torch.as_tensor(np.asarray(data['Y'], dtype=np.float32)).to(dev)
        self.n_groups, self.min_group = 0, 0
        if with_groups:
            uid = np.asarray(data['uid'], dtype=np.int64)
            order = np.argsort(uid, kind='stable')
            su = uid[order]
            starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]]) if len(su) else np.zeros(0, np.int64)
            indptr = np.r_[starts, len(su)].astype(np.int64)
            self.n_groups = len(starts)
            self.min_group = int(np.diff(indptr).min()) if self.n_groups else 0
            self.indptr = torch.as_tensor(indptr).to(dev)
            self.rows = torch.as_tensor(order.astype(np.int64)).to(dev)

    def batches(self, batch_size, dropout):
        for b0 in range(0, self.n, batch_size):
            yield {'train': False, 'rank': 1, 'dropout': dropout, 'X': self.X[b0:b0 + batch_size],
                   'Y': self.Y[b0:b0 + batch_size]}


class DeviceTrainSet(object):
    """The train interactions resident in HBM: uid/iid arrays, each user's rows (CSR, ascending sample id) and sorted
    train history (CSR).  Per epoch ONE kernel draws every row's negative (dccf_sample_train_negatives, the per-epoch
    tmp_history rule of src/data_processor/DataProcessor.py:479-517), a device permutation plays the role of
    shuffle_in_unison_scary (src/utils/utils.py:82-92), and the batches X = [pos ; neg] (:160-207) are views of one
    [n_batches, 2B, 2] tensor — no host loop, no H2D copy."""

    def __init__(self, uid, iid, user_num, item_num, seed, hist_uid=None, hist_iid=None):
        uid = np.asarray(uid, dtype=np.int64)
        iid = np.asarray(iid, dtype=np.int64)
        self.user_num, self.item_num, self.seed, self.n = int(user_num), int(item_num), int(seed), len(uid)
        order = np.argsort(uid, kind='stable')
        rows_indptr = np.searchsorted(uid[order], np.arange(self.user_num + 1)).astype(np.int64)
        hu = uid if hist_uid is None else np.asarray(hist_uid, dtype=np.int64)
        hi = iid if hist_iid is None else np.asarray(hist_iid, dtype=np.int64)
        key = np.unique(hu * self.item_num + hi)                     # sorted by user then item, de-duplicated
        ku, ki = key // self.item_num, key % self.item_num
        hist_indptr = np.searchsorted(ku, np.arange(self.user_num + 1)).astype(np.int64)
        t = utils.numpy_to_torch
        self.uid, self.iid = t(uid), t(iid)
        self.rows, self.rows_indptr = t(order.astype(np.int64)), t(rows_indptr)
        self.hist_indptr, self.hist_items = t(hist_indptr), t(ki.astype(np.int64))
        self.neg = torch.empty(self.n, dtype=torch.int64, device=self.uid.device)
        self._bad = torch.zeros(1, dtype=torch.int32, device=self.uid.device)

    def sample_negatives(self, epoch):
        return _lib.sample_train_negatives(self.rows_indptr, self.rows, self.hist_indptr, self.hist_items, self.user_num,
                                           self.item_num, self.seed, epoch, out=self.neg)

    def check_negatives(self):
        """Raises if an epoch_batches() since the last check met a user without an admissible negative (the reference asserts,
        DataProcessor.py:495).  Called at the end of an epoch: no host synchronisation inside the step loop."""
        if self.n and int(self._bad) != 0:        # (an explicit raise: an assert is stripped under python -O)
            self._bad.zero_()
            raise RuntimeError('no admissible training negative left for some user (src/data_processor/DataProcessor.py:495 asserts '
                               'here): its pairs of this epoch were trained against item 0')

    def epoch_batches(self, epoch, batch_size):
        """Two launches per epoch: the sampler and the batch tensor (the permutation — the in-unison shuffle of
        src/utils/utils.py:82-92 — is a keyed bijection of (seed, epoch) evaluated inside the batch kernel; DCCF_TORCH_PERM=1
        draws it with torch.randperm instead: a key sort of five launches)."""
        neg = self.sample_negatives(epoch)
        perm = None
        if os.environ.get('DCCF_TORCH_PERM') == '1':
            g = torch.Generator(device=self.uid.device)
            g.manual_seed((self.seed * 1000003 + int(epoch)) & 0x7FFFFFFFFFFFFFFF)
            perm = torch.randperm(self.n, generator=g, device=self.uid.device)
        # a user whose history leaves nothing to draw gets -1 from the sampler; the batch kernel stores 0 instead (an id of -1
        # must never reach a kernel as a row index) and raises the device flag check_negatives() reads
        return _lib.build_epoch_batches(self.uid, self.iid, neg, perm, batch_size, self._bad, self.seed, epoch)
