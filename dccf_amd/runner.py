# coding=utf-8
"""BaseRunner with the reference's surface (src/runners/BaseRunner.py:16-355): same flags, same epoch loop, same log
lines, same early-stop rule, same best-model save/restore.  The training step itself is the fused HIP path:
``model(batch)`` runs forward + loss + backward kernels, ``optimizer.step()`` is the dense l2 + clip + update kernel."""
import logging
import os
from time import time

import numpy as np
import pandas as pd
import torch

from dccf_amd import utils, _lib
from dccf_amd.models import FusedOptimizer


class BaseRunner(object):
    @staticmethod
    def parse_runner_args(parser):
        """src/runners/BaseRunner.py:18-48 — identical flags and defaults, plus --fused_sampling for this build."""
        parser.add_argument('--load', type=int, default=0, help='Whether load model and continue to train')
        parser.add_argument('--epoch', type=int, default=100, help='Number of epochs.')
        parser.add_argument('--check_epoch', type=int, default=1, help='Check every epochs.')
        parser.add_argument('--early_stop', type=int, default=1, help='whether to early-stop.')
        parser.add_argument('--lr', type=float, default=0.01, help='Learning rate.')
        parser.add_argument('--batch_size', type=int, default=128, help='Batch size during training.')
        parser.add_argument('--eval_batch_size', type=int, default=128 * 128, help='Batch size during testing.')
        parser.add_argument('--dropout', type=float, default=0.2, help='Dropout probability for each deep layer')
        parser.add_argument('--l2', type=float, default=1e-4, help='Weight of l2_regularize in loss.')
        parser.add_argument('--optimizer', type=str, default='GD', help='optimizer: GD, Adam, Adagrad')
        parser.add_argument('--metric', type=str, default='RMSE', help='metrics: RMSE, MAE, AUC, F1, Accuracy, Precision, Recall')
        parser.add_argument('--skip_eval', type=int, default=0, help='number of epochs without evaluation')
        parser.add_argument('--fused_sampling', type=int, default=1,
                            help='1: train negatives, batches and evaluation negatives are drawn on the GPU (Philox); '
                                 '0: the reference host path (numpy, bit-identical batches for the same seed)')
        parser.add_argument('--device_eval', type=int, default=1,
                            help='1: predictions, top-k selection and ndcg/hit/precision/recall/f1@k (k <= 1024), rmse, mae and auc stay on the '
                                 'GPU; 0: the reference host path (pandas-free numpy restatement)')
        parser.add_argument('--eval_noise', type=str, default='full',
                            help='DCCF evaluation: full = the 768-d noise of DCCF.predict, op for op; projected = its exact '
                                 'D-dimensional projection N(0, std^2 W_f W_f^T) (same output distribution, ~5x faster)')
        parser.add_argument('--overlap_opt', type=int, default=0,
                            help='1: the optimizer pass over the embedding rows a DCCF batch does not touch runs on a '
                                 'low-priority side stream beside forward/backward (dccf_train_step overlap); measured '
                                 '+5 %% at batch 128 only with DCCF_SIDE_CUS=128 and a non-default stream, hence off')
        parser.add_argument('--mp', type=str, default='replicated',
                            help='layout of a multi-GPU run (python -m torch.distributed.run --nproc-per-node G -m dccf_amd.main ...): '
                                 'replicated = every GPU holds the whole model, one all-gather of the touched gradient rows per step; '
                                 'sharded = embedding rows, optimizer state, feature rows and the rows of the exposure matrix are '
                                 'sharded by row mod G, all-to-all row exchange + all-reduce of the dense weights per step')
        parser.add_argument('--use_graph', type=int, default=0,
                            help='1: replay each DCCF training step as one hipGraph (needs --fused_sampling 1); measured '
                                 'slower than eager launches at batch 128 on MI355X (graph-launch floor), hence off')
        return parser

    def __init__(self, optimizer='GD', learning_rate=0.01, epoch=100, batch_size=128, eval_batch_size=128 * 128,
                 dropout=0.2, l2=1e-5, metrics='RMSE', check_epoch=10, early_stop=1, fused_sampling=1, use_graph=0, device_eval=1, overlap_opt=0,
                 mp='replicated'):
        if mp not in ('replicated', 'sharded'):
            raise ValueError('--mp must be replicated or sharded')
        self.mp = mp
        self.optimizer_name, self.learning_rate, self.epoch = optimizer, learning_rate, epoch
        self.batch_size, self.eval_batch_size = batch_size, eval_batch_size
        self.dropout, self.no_dropout, self.l2_weight = dropout, 0.0, l2
        self.metrics = metrics.lower().split(',')
        self.check_epoch, self.early_stop, self.fused_sampling, self.use_graph = check_epoch, early_stop, fused_sampling, use_graph
        self.device_eval, self.overlap_opt = device_eval, overlap_opt
        self.time = None
        self.train_results, self.valid_results, self.test_results = [], [], []

    def _build_optimizer(self, model):
        """src/runners/BaseRunner.py:83-107."""
        name = self.optimizer_name.lower()
        if name not in ('gd', 'adagrad', 'adam'):
            logging.error('Unknown Optimizer: ' + self.optimizer_name)
            assert self.optimizer_name in ['GD', 'Adagrad', 'Adam']
        logging.info('Optimizer: ' + {'gd': 'GD', 'adagrad': 'Adagrad', 'adam': 'Adam'}[name])
        return FusedOptimizer(model, name, self.learning_rate, self.l2_weight, clip=50.0)

    def _check_time(self, start=False):
        if self.time is None or start:
            self.time = [time()] * 2
            return self.time[0]
        t = self.time[1]
        self.time[1] = time()
        return self.time[1] - t

    def batches_add_control(self, batches, train):
        for b in batches:
            b['train'] = train
            b['dropout'] = self.dropout if train else self.no_dropout
        return batches

    def predict(self, model, data, data_processor):
        """src/runners/BaseRunner.py:134-157: batched predict, reordered by sample_id."""
        batches = self.batches_add_control(data_processor.prepare_batches(data, self.eval_batch_size, train=False), train=False)
        model.eval()
        if hasattr(model, 'begin_eval'):
            model.begin_eval()
        preds = [model.predict(b)['prediction'] for b in batches]
        predictions = torch.cat(preds).cpu().numpy() if preds else np.zeros(0, dtype=np.float32)
        sample_ids = np.concatenate([b[utils.K_SAMPLE_ID] for b in batches])
        out = np.empty_like(predictions)
        # reorder_dict of the reference: later duplicates win; sample ids of one split are unique
        pos = np.empty(int(sample_ids.max()) + 1 if len(sample_ids) else 0, dtype=np.int64)
        pos[sample_ids] = np.arange(len(sample_ids))
        out = predictions[pos[data[utils.K_SAMPLE_ID]]]
        return out

    def fit(self, model, data, data_processor, epoch=-1):
        """src/runners/BaseRunner.py:159-191.  One optimizer step per batch (accumulate_size reaches the threshold on
        every batch in the reference, :172-188)."""
        if model.optimizer is None:
            model.optimizer = self._build_optimizer(model)
        if utils.world_size() > 1:
            if self.mp == 'sharded':
                return self._fit_sharded(model, data_processor, epoch)
            return self._fit_replicated(model, data_processor, epoch)
        model.train()
        out = None
        if self.fused_sampling and data_processor.rank == 1:
            full, tail = data_processor.device_epoch(max(epoch, 0), self.batch_size)
            B = self.batch_size
            if self.use_graph and getattr(model, 'kind', '') == 'DCCF' and full.shape[0] > 0:
                # the whole step (5 kernels + dense optimizer) is one hipGraph replay per batch
                sg = getattr(model, '_step_graph', None)
                if sg is None or sg.nb != full.shape[0] or sg.rows != 2 * B:
                    from dccf_amd.models import StepGraph
                    sg = model._step_graph = StepGraph(model, model.optimizer, full.shape[0], 2 * B, self.dropout)
                sg.load_epoch(full)
                for _ in range(full.shape[0]):
                    sg.step()
                out = sg.out_dict()
                if tail is not None:
                    pred = sg.tail(tail)
                    out = {'prediction': pred, 'check': [('prediction', pred)], 'loss': model._loss[0]}
                data_processor._dev.check_negatives()
                model.eval()
                return out
            y = torch.cat([torch.ones(B, device=full.device), torch.zeros(B, device=full.device)])
            batch = {'Y': y, 'rank': 1, 'train': True, 'dropout': self.dropout, utils.REAL_BATCH_SIZE: B}
            nb = full.shape[0]
            for k in range(nb):
                batch['X'] = full[k]
                # the next batch is known: its candidates are drawn inside this step's optimizer launch
                out = self._step(model, batch, X_next=full[k + 1] if k + 1 < nb else None)
            if tail is not None:
                r = tail.shape[0] // 2
                batch = {'X': tail, 'Y': torch.cat([y[:r], y[B:B + r]]), 'rank': 1, 'train': True, 'dropout': self.dropout,
                         utils.REAL_BATCH_SIZE: r}
                out = self._step(model, batch)
            data_processor._dev.check_negatives()
        else:
            batches = self.batches_add_control(data_processor.prepare_batches(data, self.batch_size, train=True), train=True)
            for batch in batches:
                out = self._step(model, batch)
        model.eval()
        return out

    def _fit_replicated(self, model, data_processor, epoch):
        """One epoch on the G GPUs of a node (launched with torch.distributed.run, one process per GPU; no reference counterpart:
        src/main.py:106,153-155 is single-GPU).  Every rank holds the whole model and the whole train set and draws the SAME
        epoch (negatives, permutation: counter-based in (seed, epoch)); optimizer step j trains the G consecutive batches
        j G .. j G + G - 1 of the epoch's schedule, rank r the r-th of them, as ONE step on their union — the reference's
        step at batch size G x --batch_size (its loss is a sum, src/models/DCCF.py:116-120) — through dccf_amd.replicated
        on THIS model's parameter buffer and this optimizer's state.  What is left when the batches do not divide by G
        (< G batches + the epoch's short last batch) is one more step on equal shares per rank, the shares completed with the
        epoch's first pairs (at most G - 1 of them: what DistributedSampler does).  The replicas stay bit-identical, so every
        rank evaluates for itself and takes the same early-stopping decisions; rank 0 alone writes files."""
        import torch.distributed as dist
        from dccf_amd import replicated
        G, rank, B = dist.get_world_size(), dist.get_rank(), self.batch_size
        o = model.optimizer
        tr = getattr(model, '_replicated', None)
        if tr is None:
            D = getattr(model, 'ui_vector_size', 0)
            if (getattr(model, 'kind', '') != 'DCCF' or D % 4 != 0 or not 4 <= D <= 128 or not self.fused_sampling
                    or data_processor.rank != 1):
                raise RuntimeError('training on several GPUs (replicated layout) covers --model_name DCCF --rank 1 --fused_sampling 1 '
                                   'with an embedding size that is a multiple of 4 up to 128 (--mp sharded: up to 256)')
            o.flush()
            tr = replicated.ReplicatedDCCF(rank, G, model.user_num, model.item_num, D, model.sample_num, model.attribute_num,
                                           model.std, self.dropout, o.lr, o.l2, model.random_seed,
                                           replicated.HipBackend(model.device), model.device, model.feature_embedding,
                                           expo=model.expo_prob, ips=model.ips_factors, max_rows=2 * B, overlap=True,
                                           flat_p=model.flat_p, s1=o.s1, s2=o.s2, opt_name=o.name, n_layers=model.n_layers)
            tr.t = o.t
            model._replicated = tr
            o.lazy = None                   # the trainer owns the lazy regularisation of these buffers from here on
            _flush0 = o.flush
            o.flush = lambda: (tr.flush(), _flush0())[1]          # evaluation / checkpoints read every row
        model.train()
        full, tail = data_processor.device_epoch(max(epoch, 0), B)
        dev = full.device
        sched, last = replicated.epoch_schedule(full, tail, G)
        ns = sched.shape[0]
        y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
        pred = torch.empty(2 * B, dtype=torch.float32, device=dev)
        # no Philox word twice: the trainer's words of this epoch start after everything the model's call counter handed out
        tr.word_base = model._call + 1 - tr.t * G
        loss = None
        # first contact with the peers (the first collectives of the job) under a host-side deadline: a hang exits non-zero
        first = not getattr(tr, '_met_peers', False)
        dl = utils.Deadline(float(os.environ.get('DCCF_WARMUP_DEADLINE_S', '420')) if first else 0, 'the first replicated multi-rank step')
        for j in range(ns):
            _, loss = tr.train_step(sched[j, rank], y, pred, X_all=sched[j], X_all_next=sched[j + 1] if j + 1 < ns else None)
            if j == 0 and first:
                torch.cuda.synchronize()
                dl.cancel()
        if last is not None:         # the rest of the epoch: one step of equal shares
            b = last.shape[1] // 2
            yl = torch.cat([torch.ones(b, device=dev), torch.zeros(b, device=dev)])
            pred = torch.empty(2 * b, dtype=torch.float32, device=dev)
            _, loss = tr.train_step(last[rank], yl, pred, X_all=last)
        dl.cancel()
        tr._met_peers = True
        tr.flush()
        if not tr.crosscheck_replicas():        # (stderr has the details; every rank resynchronised from rank 0)
            logging.warning('replicas differed after epoch %d: resynchronised from rank 0, synchronous step from here on' % epoch)
        o.t = tr.t
        model._call = tr.word_base + tr.t * G        # every word below is used
        data_processor._dev.check_negatives()
        model.eval()
        return {'prediction': pred, 'check': [('prediction', pred)], 'loss': loss[0] if loss is not None else model._loss[0]}

    def _fit_sharded(self, model, data_processor, epoch):
        """One epoch on the G GPUs of a node in the ROW-SHARDED layout (--mp sharded; dccf_amd/sharded.py; no reference
        counterpart: src/main.py:106,153-155 is single-GPU).  Rank r trains with — and holds the optimizer state of — the embedding
        rows r, r + G, ...; the feature rows and the rows of the exposure matrix (or the IPS factors) of those rows live with
        them; W, b and the extra layers are replicated.  Same schedule as the replicated layout: optimizer step j trains batches
        j G .. j G + G - 1 of the epoch as ONE step on their union (the reference's step at batch size G x --batch_size), what
        does not divide is one more step of equal shares.  GD / Adagrad / Adam, --n_layers 1 .. 8, any embedding size up to 128.
        After the epoch every rank gathers the tables into its model (one all-gather per table) and evaluates for itself — the
        gathered replicas are identical, so the ranks take the same checkpoint / early-stopping decisions; rank 0 writes files."""
        import torch.distributed as dist
        from dccf_amd import sharded, replicated
        G, rank, B = dist.get_world_size(), dist.get_rank(), self.batch_size
        o = model.optimizer
        p = model.params
        names = ['uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias']
        tr = getattr(model, '_sharded', None)
        first = tr is None
        if first:
            if getattr(model, 'kind', '') != 'DCCF' or not self.fused_sampling or data_processor.rank != 1:
                raise RuntimeError('training on several GPUs covers --model_name DCCF --rank 1 --fused_sampling 1')
            o.flush()
            dev = model.device
            ips_local, expo_local = None, None
            if model.expo_prob is not None:          # the dense <ds>.ips_expo_prob.npy: this rank keeps the rows of its users
                expo_local = model.expo_prob[rank::G].contiguous()
            else:
                f = model.ips_factors
                ips_local = dict(P=f['P'][rank::G].contiguous(), bu=f['bu'][rank::G].contiguous(), Q=f['Q'][rank::G].contiguous(),
                                 bi=f['bi'][rank::G].contiguous(), prop=f['prop'][rank::G].contiguous(), b0=f['b0'], M=f['M'])
            tr = sharded.ShardedDCCF(rank, G, model.user_num, model.item_num, model.ui_vector_size, model.sample_num,
                                     model.attribute_num, model.std, self.dropout, o.lr, o.l2, model.random_seed,
                                     sharded.HipBackend(dev), dev, model.feature_embedding[rank::G].contiguous(), ips_local,
                                     expo_local=expo_local, opt_name=o.name, n_layers=model.n_layers)
            tr.set_global_params(p[names[0]], p[names[1]], p[names[2]], p[names[3]], extra=model._extra(p))
            for full, mine in ((o.s1, tr.s1), (o.s2, tr.s2)):        # the optimizer's state so far (zeros in a fresh run)
                if full is None:
                    continue
                mv, tv = model.views_of(full), tr.views_of(mine)
                tv[0].copy_(mv[names[0]][rank::G])
                tv[1].copy_(mv[names[1]][rank::G])
                for dst, name in zip(tv[2:], list(mv)[2:]):
                    dst.copy_(mv[name])
            tr.t = o.t
            logging.info('# row-sharded training on %d ranks, collectives: %s' % (G, tr.crosscheck_collectives()))
            model._sharded = tr
            o.lazy = None                   # the trainer owns the parameters between the gathers
        model.train()
        full, tail = data_processor.device_epoch(max(epoch, 0), B)
        dev = full.device
        sched, last = replicated.epoch_schedule(full, tail, G)
        ns = sched.shape[0]
        # no Philox word twice: two words for the candidate streams of the epoch's two plans, then G words per optimizer step,
        # all after everything the model's call counter handed out (evaluation passes draw from it between the epochs)
        c0 = model._call + 1
        tr.word_base = c0 + 2 - tr.t * G
        pred = loss = None
        # first contact with the peers (the first collectives of the job) under a host-side deadline: a hang exits non-zero
        dl = utils.Deadline(float(os.environ.get('DCCF_WARMUP_DEADLINE_S', '420')) if first else 0, 'the first row-sharded step')
        if ns > 0:
            tr.begin_epoch(sched, c0)
            for k in range(ns):
                pred, loss = tr.train_step(k)
                if k == 0 and first:
                    torch.cuda.synchronize()
                    dl.cancel()
        if last is not None:         # the rest of the epoch: one step of equal shares
            tr.begin_epoch(last.unsqueeze(0).contiguous(), c0 + 1)
            pred, loss = tr.train_step(0)
        dl.cancel()
        tr.flush()
        o.t = tr.t
        model._call = tr.word_base + tr.t * G        # every word below is used
        # every rank's model gets the whole tables back (evaluation, checkpoints, l2): W, b and the extra layers are replicated
        tr.gather_tables(p[names[0]], p[names[1]])
        p[names[2]].copy_(tr.W)
        p[names[3]].copy_(tr.b)
        for (w, bb), (w0, b0) in zip(model._extra(p), tr.extra):
            w.copy_(w0)
            bb.copy_(b0)
        data_processor._dev.check_negatives()
        model.eval()
        return {'prediction': pred, 'check': [('prediction', pred)], 'loss': loss[0] if loss is not None else model._loss[0]}

    def _step(self, model, batch, X_next=None):
        """The body of the reference's batch loop (src/runners/BaseRunner.py:172-188)."""
        if hasattr(model, 'train_step'):
            return model.train_step(batch, overlap=self.overlap_opt, X_next=X_next)    # one library call per step
        model.optimizer.zero_grad()
        out = model(batch)
        model.optimizer.step()        # + l2 term, clip_grad_value_(50), update: one dense kernel
        return out

    def eva_termination(self, model):
        """src/runners/BaseRunner.py:193-210."""
        metric, valid = self.metrics[0], self.valid_results
        if len(valid) > 20 and metric in utils.LOWER_METRIC_LIST and utils.strictly_increasing(valid[-5:]):
            return True
        elif len(valid) > 20 and metric not in utils.LOWER_METRIC_LIST and utils.strictly_decreasing(valid[-5:]):
            return True
        elif len(valid) - valid.index(utils.best_result(metric, valid)) > 20:
            return True
        return False

    def train(self, model, data_processor, skip_eval=0):
        """src/runners/BaseRunner.py:212-303."""
        train_data = data_processor.get_train_data(epoch=-1)
        validation_data = data_processor.get_validation_data()
        test_data = data_processor.get_test_data()
        self._check_time(start=True)
        nm = [-1.0] * len(self.metrics)
        init_train = self.evaluate(model, train_data, data_processor, metrics=['rmse', 'mae']) if train_data is not None else nm
        init_valid = self.evaluate(model, validation_data, data_processor) if validation_data is not None else nm
        init_test = self.evaluate(model, test_data, data_processor) if test_data is not None else nm
        self.init_results = (init_train, init_valid, init_test)
        logging.info('Init: \t train= %s validation= %s test= %s [%.1f s] ' % (
            utils.format_metric(init_train), utils.format_metric(init_valid), utils.format_metric(init_test),
            self._check_time()) + ','.join(self.metrics))
        try:
            for epoch in range(self.epoch):
                self._check_time()
                # the host-side shuffle of the train dict (src/runners/BaseRunner.py:246, utils.py:82-92) only feeds the host
                # batch path; with fused sampling the epoch's permutation is drawn on the GPU (DeviceTrainSet)
                fused = self.fused_sampling and data_processor.rank == 1
                epoch_train_data = train_data if fused else data_processor.get_train_data(epoch=epoch)
                last_batch = self.fit(model, epoch_train_data, data_processor, epoch=epoch)
                if self.check_epoch > 0 and (epoch == 1 or epoch % self.check_epoch == 0):
                    self.check(model, last_batch)
                training_time = self._check_time()
                if epoch >= skip_eval:
                    train_result = self.evaluate(model, train_data, data_processor, metrics=['rmse', 'mae']) if train_data is not None else nm
                    valid_result = self.evaluate(model, validation_data, data_processor) if validation_data is not None else nm
                    test_result = self.evaluate(model, test_data, data_processor) if test_data is not None else nm
                    testing_time = self._check_time()
                    self.train_results.append(train_result)
                    self.valid_results.append(valid_result)
                    self.test_results.append(test_result)
                    logging.info('Epoch %5d [%.1f s]\t train= %s validation= %s test= %s [%.1f s] '
                                 % (epoch + 1, training_time, utils.format_metric(train_result),
                                    utils.format_metric(valid_result), utils.format_metric(test_result), testing_time)
                                 + ','.join(self.metrics))
                    if utils.best_result(self.metrics[0], self.valid_results) == self.valid_results[-1]:
                        model.save_model()
                    if self.eva_termination(model) and self.early_stop == 1:
                        logging.info('Early stop at %d based on validation result.' % (epoch + 1))
                        break
                if epoch < skip_eval:
                    logging.info('Epoch %5d [%.1f s]' % (epoch + 1, training_time))
        except KeyboardInterrupt:
            logging.info('Early stop manually')
        if self.valid_results:
            for what, res in (('validation', self.valid_results), ('test', self.test_results)):
                best = utils.best_result(self.metrics[0], res)
                be = res.index(best)
                logging.info('Best Iter(%s)= %5d\t train= %s valid= %s test= %s [%.1f s] '
                             % (what, be + 1, utils.format_metric(self.train_results[be]),
                                utils.format_metric(self.valid_results[be]), utils.format_metric(self.test_results[be]),
                                self.time[1] - self.time[0]) + ','.join(self.metrics))
            model.load_model()

    @staticmethod
    def _device_metrics_ok(metrics):
        ks = set()
        for m in metrics:
            if m in ('rmse', 'mae', 'auc'):
                continue
            name, _, k = m.partition('@')
            if name not in ('ndcg', 'hit', 'precision', 'recall', 'f1') or not k.isdigit() or not 1 <= int(k) <= 1024:
                return False
            ks.add(int(k))
        return len(ks) <= 4

    @staticmethod
    def _device_auc(p, y):
        """roc_auc_score(l, p) of src/models/BaseModel.py:72-73 without leaving the GPU: the area under the ROC curve is the
        Mann-Whitney statistic with tied scores counted half — (sum of the positives' average ranks - n_pos (n_pos + 1) / 2) /
        (n_pos n_neg) — one device sort of the predictions; twice the average rank is an integer, so the sum is exact."""
        pos = y > 0
        n_pos = int(pos.sum())
        n_neg = int(y.numel()) - n_pos
        if n_pos == 0 or n_neg == 0:
            raise ValueError('Only one class present in y_true. ROC AUC score is not defined in that case.')   # sklearn's
        sp, order = torch.sort(p)
        _, inv, cnt = torch.unique_consecutive(sp, return_inverse=True, return_counts=True)
        twice_rank = 2 * torch.cumsum(cnt, 0) - cnt + 1          # 2 x the average 1-based rank of each tie group
        s2 = int(twice_rank[inv][pos[order]].sum())
        return (s2 - n_pos * (n_pos + 1)) / (2.0 * n_pos * n_neg)

    def predict_device(self, model, data, data_processor):
        """BaseRunner.predict (:134-157) from the resident split: predictions in sample-id order, on the GPU."""
        es = data_processor.device_eval_set(data)
        model.eval()
        if hasattr(model, 'begin_eval'):
            model.begin_eval()
        world, rank = utils.dist_world_rank()
        if world > 1 and os.environ.get('DCCF_SHARD_EVAL', '1') != '0' and hasattr(model, '_call'):
            # Under a multi-rank launch the replicas hold identical parameters: the evaluation batches are dealt to the ranks round
            # robin and ONE sum-all-reduce (every other rank contributes zeros: exact) gives every rank every prediction.  Batch b
            # draws with the Philox step the single-process loop would have given it (the model's call counter), so the predictions —
            # and the metrics every rank then computes for itself — are bit for bit those of the unsharded pass.
            import torch.distributed as dist
            p = torch.zeros(es.n, dtype=torch.float32, device=es.Y.device)
            c0, nb, bs = model._call, 0, self.eval_batch_size
            for b, batch in enumerate(es.batches(bs, self.no_dropout)):
                nb = b + 1
                if b % world != rank:
                    continue
                model._call = c0 + b
                p[b * bs:b * bs + batch['X'].shape[0]] = model.predict(batch)['prediction']
            model._call = c0 + nb
            if es.n:
                dist.all_reduce(p)
            return p
        preds = [model.predict(b)['prediction'] for b in es.batches(self.eval_batch_size, self.no_dropout)]
        return torch.cat(preds) if preds else torch.zeros(0, dtype=torch.float32, device=es.Y.device)

    def evaluate_device(self, model, data, data_processor, metrics, return_predictions=False):
        """evaluate() without leaving the GPU: batched predict straight from the resident split, then one wave per user
        selects the top-k and scores it (rank_eval_topk) — replaces BaseRunner.py:134-157 + BaseModel.py:55-128."""
        es = data_processor.device_eval_set(data)
        p = self.predict_device(model, data, data_processor)
        ks = sorted({int(m.split('@')[1]) for m in metrics if '@' in m})
        per_user = None
        if ks:
            if any(m.startswith('precision@') for m in metrics) and es.min_group < max(
                    int(m.split('@')[1]) for m in metrics if m.startswith('precision@')):
                raise ValueError('Relevance score length < k')        # rank_metrics.py:80-81
            per_user = _lib.rank_eval_topk(p, es.Y, es.indptr, es.rows, ks).double()
        out = []
        for m in metrics:
            if m == 'rmse':
                out.append(float(torch.sqrt(torch.mean((es.Y.double() - p.double()) ** 2))))
            elif m == 'mae':
                out.append(float(torch.mean(torch.abs(es.Y.double() - p.double()))))
            elif m == 'auc':
                out.append(self._device_auc(p, es.Y))
            else:
                name, k = m.split('@')
                j, k = ks.index(int(k)), int(k)
                col = {'ndcg': 0, 'hit': 1, 'precision': 2, 'recall': 3}.get(name)
                if col is not None:
                    vals = per_user[:, j, col]
                else:       # f1@k = 2 * hits / (k + positives)
                    vals = 2.0 * per_user[:, j, 2] * k / (k + per_user[:, len(ks), 0])
                out.append(float(vals.mean()))
        return (out, p) if return_predictions else out

    def evaluate(self, model, data, data_processor, metrics=None, write_rank=False):
        """src/runners/BaseRunner.py:305-332."""
        if metrics is None:
            metrics = self.metrics
        if self.device_eval and self._device_metrics_ok(metrics):
            if not write_rank:
                return self.evaluate_device(model, data, data_processor, metrics)
            res, p = self.evaluate_device(model, data, data_processor, metrics, return_predictions=True)
            self._write_rank(os.path.join(data_processor.data_loader.path, utils.RANK_FILE_NAME), data, p.cpu().numpy())
            return res
        predictions = self.predict(model, data, data_processor)
        if write_rank:
            self._write_rank(os.path.join(data_processor.data_loader.path, utils.RANK_FILE_NAME), data, predictions)
        return model.evaluate_method(predictions, data, metrics=metrics)

    @staticmethod
    def _write_rank(path, data, predictions):
        """rank.csv of src/runners/BaseRunner.py:315-323: tab-separated uid, iid, score, label sorted by uid.  With
        --test_neg_n 1000 the test split of an Electronics-size dataset is ~2e8 rows: pandas needs minutes for the sort
        and the text conversion, so large frames go through a stable numpy argsort and pyarrow's CSV writer."""
        if not utils.is_rank0():
            return
        n = len(predictions)
        if n >= 2000000:
            try:
                import pyarrow as pa
                import pyarrow.csv as pacsv
                order = np.argsort(np.asarray(data['uid']), kind='stable')
                label = np.asarray(data['Y'])[order]
                table = pa.table({'uid': np.asarray(data['uid'])[order], 'iid': np.asarray(data['iid'])[order],
                                  'score': np.asarray(predictions, dtype=np.float32)[order],
                                  'label': pa.array(label, type=pa.float32()).cast(pa.float64())})
                with open(path, 'wb') as f:            # pandas' header (unquoted)
                    f.write(b'uid\tiid\tscore\tlabel\n')
                    pacsv.write_csv(table, f, pacsv.WriteOptions(delimiter='\t', quoting_style='none', include_header=False))
                return
            except ImportError:
                pass
        df = pd.DataFrame({'uid': data['uid'], 'iid': data['iid'], 'score': predictions, 'label': data['Y']})
        df = df.sort_values(by='uid')
        df.to_csv(path, sep='\t', index=False)

    def check(self, model, out_dict):
        """src/runners/BaseRunner.py:334-355."""
        logging.info(os.linesep)
        for name, t in out_dict['check']:
            d = np.array(t.detach().cpu())
            logging.info(os.linesep.join([name + '\t' + str(d.shape), np.array2string(d, threshold=20)]) + os.linesep)
        loss, l2 = float(out_dict['loss']), float(model.l2()) * self.l2_weight
        logging.info('loss = %.4f, l2 = %.4f' % (loss, l2))
        if not (abs(loss) * 0.005 < l2 < abs(loss) * 0.1):
            logging.warning('l2 inappropriate: loss = %.4f, l2 = %.4f' % (loss, l2))
