# coding=utf-8
"""DataLoader with the reference's surface (src/data_loaders/DataLoader.py:13-290): same flags, same on-disk formats
(header-less ``uid,iid,label,time`` CSVs, ``<ds>.info.json``, ``<ds>.train_group.csv``, ``<ds>.vt_group.csv``), same
attributes.  Out of scope for acceleration (runs once, I/O bound — SURVEY.md §2.1 row 9)."""
import json
import logging
import os

import numpy as np
import pandas as pd

from dccf_amd import utils


def group_user_interactions_df(in_df, label='label', seq_sep=','):
    """src/utils/mining.py:18-29: one row per user, positives joined by seq_sep (uid ascending, file order inside).
    Vectorised: a stable sort by uid instead of the reference's per-group Python loop (same rows, same order)."""
    df = in_df[in_df[label] > 0] if label in in_df.columns else in_df
    uid = df['uid'].values
    order = np.argsort(uid, kind='stable')
    su, si = uid[order], df['iid'].values[order].astype(str)
    starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]]) if len(su) else np.zeros(0, np.int64)
    ends = np.r_[starts[1:], len(su)]
    out = pd.DataFrame()
    out['uid'] = su[starts]
    out['iids'] = [seq_sep.join(si[a:b]) for a, b in zip(starts, ends)]
    return out


class DataLoader(object):
    @staticmethod
    def parse_data_args(parser):
        parser.add_argument('--path', type=str, default='../datasets/', help='Input data dir.')
        parser.add_argument('--dataset', type=str, default='ml100k-1-5', help='Choose a dataset.')
        parser.add_argument('--sep', type=str, default=',', help='sep of csv file.')
        parser.add_argument('--label', type=str, default='label', help='name of dataset label column.')
        return parser

    def __init__(self, path, dataset, label='label', load_data=True, sep='\t', seqs_sep=','):
        self.dataset = dataset
        self.path = os.path.join(path, dataset)
        base = os.path.join(self.path, dataset)
        self.train_file, self.validation_file, self.test_file = (base + utils.TRAIN_SUFFIX, base + utils.VALIDATION_SUFFIX,
                                                                 base + utils.TEST_SUFFIX)
        self.info_file = base + utils.INFO_SUFFIX
        self.user_file, self.item_file = base + utils.USER_SUFFIX, base + utils.ITEM_SUFFIX
        self.train_his_file, self.vt_his_file = base + utils.TRAIN_GROUP_SUFFIX, base + utils.VT_GROUP_SUFFIX
        self.sep, self.seqs_sep, self.load_data, self.label = sep, seqs_sep, load_data, label
        self.train_df, self.validation_df, self.test_df = None, None, None
        self.user_df, self.item_df = None, None
        if load_data and os.path.exists(self.user_file):
            self.user_df = pd.read_csv(self.user_file, sep='\t')
        if load_data and os.path.exists(self.item_file):
            self.item_df = pd.read_csv(self.item_file, sep='\t')
        self._load_data()
        self._load_his()
        self._load_info()

    def _load_data(self):
        names = ['uid', 'iid', 'label', 'time']
        for attr, f, what in (('train_df', self.train_file, 'train'), ('validation_df', self.validation_file, 'validation'),
                              ('test_df', self.test_file, 'test')):
            if os.path.exists(f) and self.load_data:
                logging.info('load %s csv...' % what)
                setattr(self, attr, pd.read_csv(f, sep=self.sep, names=names))
                logging.info('size of %s: %d' % (what, len(getattr(self, attr))))

    def _load_info(self):
        if not os.path.exists(self.info_file):
            mx, mn = {}, {}
            for df in (self.train_df, self.validation_df, self.test_df, self.user_df, self.item_df):
                if df is None:
                    continue
                for c in df.columns:
                    mx[c] = int(df[c].max()) if c not in mx else max(int(df[c].max()), mx[c])
                    mn[c] = int(df[c].min()) if c not in mn else min(int(df[c].min()), mn[c])
            with open(self.info_file, 'w') as f:
                f.write(json.dumps(mx) + os.linesep + json.dumps(mn))
        else:
            lines = open(self.info_file, 'r').readlines()
            mx, mn = json.loads(lines[0]), json.loads(lines[1])
        self.column_max, self.column_min = mx, mn
        self.label_max, self.label_min = mx[self.label], mn[self.label]
        logging.info('label: %d-%d' % (self.label_min, self.label_max))
        self.user_num = mx['uid'] + 1 if 'uid' in mx else 0
        self.item_num = mx['iid'] + 1 if 'iid' in mx else 0
        logging.info('# of users: %d' % self.user_num)
        logging.info('# of items: %d' % self.item_num)
        self.user_features = [f for f in mx if f.startswith('u_')]
        self.item_features = [f for f in mx if f.startswith('i_')]
        self.context_features = [f for f in mx if f.startswith('c_')]
        self.features = self.context_features + self.user_features + self.item_features
        logging.info('# of features: %d' % len(self.features))

    def _load_his(self):
        if not self.load_data:
            return
        if not os.path.exists(self.train_his_file):
            logging.info('building train history csv...')
            group_user_interactions_df(self.train_df, self.label, self.seqs_sep).to_csv(self.train_his_file, index=False, sep=self.sep)
        if not os.path.exists(self.vt_his_file):
            logging.info('building vt history csv...')
            vt = pd.concat([self.validation_df, self.test_df])
            group_user_interactions_df(vt, self.label, self.seqs_sep).to_csv(self.vt_his_file, index=False, sep=self.sep)

        def build(df):
            return dict(zip(df['uid'].tolist(), [[int(j) for j in s.split(self.seqs_sep)] for s in df['iids'].astype(str)]))

        logging.info('load history csv...')
        self.train_his_df = pd.read_csv(self.train_his_file, sep=self.sep)
        self.train_user_his = build(self.train_his_df)
        self.vt_his_df = pd.read_csv(self.vt_his_file, sep=self.sep)
        self.vt_user_his = build(self.vt_his_df)

    def feature_info(self, include_id=True, include_item_features=True, include_user_features=True):
        """src/data_loaders/DataLoader.py:196-224."""
        features = []
        if include_id:
            features.extend(['uid', 'iid'])
        if include_user_features:
            features.extend(self.user_features)
        if include_item_features:
            features.extend(self.item_features)
        dims, fmin, fmax = 0, [], []
        for f in features:
            fmin.append(dims)
            dims += int(self.column_max[f] + 1)
            fmax.append(dims - 1)
        logging.info('Model # of features %d' % len(features))
        logging.info('Model # of feature dims %d' % dims)
        return features, dims, fmin, fmax

    def drop_neg(self):
        """src/data_loaders/DataLoader.py:276-290: top-n task keeps label > 0 and sets it to 1."""
        logging.info('Drop Neg Samples...')
        for attr in ('train_df', 'validation_df', 'test_df'):
            df = getattr(self, attr)
            df = df[df[self.label] > 0].reset_index(drop=True)
            df[self.label] = 1
            setattr(self, attr, df)
            logging.info('size of %s: %d' % (attr[:-3], len(df)))
