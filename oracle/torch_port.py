# coding=utf-8
"""ORACLE — test infrastructure, NOT product code.

A PyTorch-CPU restatement of the reference's DCCF training step (src/models/DCCF.py:66-127 +
src/runners/BaseRunner.py:178-187), written from scratch, used for two things only:
  * ``bench.py``'s ``cpu_baseline`` leg ("port"): the reference's op sequence (embedding gathers of the 22x / 2x
    duplicated rows, a fresh [L,768] normal_ per step, cat + Linear + relu + dropout, softmax over the candidates,
    BPR, dense l2 over all parameters, autograd backward, clip_grad_value_(50), torch.optim.Adam with weight_decay)
    timed on the GPU box's host cores, because /root/reference cannot travel there;
  * a second, autograd-based check of the analytic backward in oracle/dccf_oracle.py (tests/test_oracle_golden.py).
"""
import numpy as np
import torch


class DCCFPort(torch.nn.Module):
    def __init__(self, user_num, item_num, D, feat, expo_rows, S=10, A=2, std=0.1, seed=2019):
        """feat: [I,F] tensor; expo_rows: [U_e, I] tensor holding the exposure rows of users < U_e (the bounded CPU
        sample only touches those users)."""
        super().__init__()
        torch.manual_seed(seed)
        self.uid_embeddings = torch.nn.Embedding(user_num, D)
        self.iid_embeddings = torch.nn.Embedding(item_num, D)
        self.mlp = torch.nn.ModuleList([torch.nn.Linear(D + feat.shape[1], D)])
        self.feature_embedding, self.expo_prob = feat, expo_rows
        self.item_num, self.S, self.A, self.std = item_num, S, A, std
        for m in self.modules():          # BaseModel.init_paras (src/models/BaseModel.py:130-142)
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.normal_(m.weight, 0.0, 0.01)
                torch.nn.init.normal_(m.bias, 0.0, 0.01)
            elif isinstance(m, torch.nn.Embedding):
                torch.nn.init.normal_(m.weight, 0.0, 0.01)

    def predict(self, X, dropout, sample_item=None, noise=None, keep=None):
        u_ids, i_ids = X[:, 0], X[:, 1]
        N = X.shape[0]
        if sample_item is None:
            sample_item = torch.randint(self.item_num, size=(N, self.S))
        items = torch.cat((i_ids.view(-1, 1), sample_item), 1)
        items = items.view(-1, self.S + 1, 1).expand(N, self.S + 1, self.A)
        users = u_ids.view(-1, 1, 1).expand(N, self.S + 1, self.A)
        true_items = i_ids.view(-1, 1, 1).expand(N, self.S + 1, self.A)
        uid, iid, fid = users.reshape(-1), items.reshape(-1), true_items.reshape(-1)
        ue, ie = self.uid_embeddings(uid), self.iid_embeddings(iid)
        fe = self.feature_embedding[fid]
        if noise is None:
            noise = torch.empty(fe.shape).normal_(std=self.std)
        x = torch.cat((ie, fe + noise), 1)
        h = torch.relu(self.mlp[0](x))
        if keep is not None:
            h = h * keep.div(1.0 - dropout)
        elif dropout > 0:
            h = torch.nn.functional.dropout(h, p=dropout, training=True)
        m = (ue * h).sum(dim=1).reshape(N, self.S + 1, self.A)
        w = torch.softmax(self.expo_prob[uid, iid].reshape(N, self.S + 1, self.A), dim=1)
        return (w * m).sum(dim=1).mean(dim=1).view(-1)

    def l2(self):
        s = 0
        for p in self.parameters():
            s = s + (p ** 2).sum()
        return s


def train_steps(model, X_batches, dropout=0.2, lr=1e-3, l2=1e-4):
    """The body of BaseRunner.fit for rank==1 batches ([pos ; neg] rows).  Returns the losses."""
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=l2)
    losses = []
    for X in X_batches:
        opt.zero_grad()
        pred = model.predict(X, dropout)
        B = X.shape[0] // 2
        loss = -(pred[:B] - pred[B:]).sigmoid().log().sum()
        total = loss + model.l2() * l2
        total.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 50)
        opt.step()
        losses.append(float(loss))
    return losses


def synthetic_batches(user_num_sample, item_num, batch_size, n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        u = rng.randint(0, user_num_sample, batch_size)
        X = np.concatenate([np.stack([u, rng.randint(0, item_num, batch_size)], 1),
                            np.stack([u, rng.randint(0, item_num, batch_size)], 1)]).astype(np.int64)
        out.append(torch.from_numpy(X))
    return out
