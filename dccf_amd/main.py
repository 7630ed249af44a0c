# coding=utf-8
"""Command line with the reference's surface (src/main.py:24-195):

    python -m dccf_amd.main --rank 1 --model_name DCCF --optimizer Adam --lr 0.001 --dataset Electronics \
        --metric ndcg@5,recall@5,precision@5 --gpu 0 --epoch 100 --test_neg_n 1000

Two-phase argparse, the same flags contributed by the same classes, the same log / model / result file naming.
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

from dccf_amd import utils
from dccf_amd.data_loader import DataLoader
from dccf_amd.data_processor import DataProcessor
from dccf_amd.models import BaseModel, RecModel, BiasedMF, IPSBiasedMF, DCCF
from dccf_amd.runner import BaseRunner

CLASSES = {'DataLoader': DataLoader, 'DataProcessor': DataProcessor, 'BaseRunner': BaseRunner, 'BaseModel': BaseModel,
           'RecModel': RecModel, 'BiasedMF': BiasedMF, 'IPSBiasedMF': IPSBiasedMF, 'DCCF': DCCF}


def build_model(model_name, model_cls, args, data_loader):
    """The per-name constructor branches of src/main.py:120-148."""
    common = dict(label_min=data_loader.label_min, label_max=data_loader.label_max, feature_num=0,
                  user_num=data_loader.user_num, item_num=data_loader.item_num, u_vector_size=args.u_vector_size,
                  i_vector_size=args.i_vector_size, random_seed=args.random_seed, model_path=args.model_path)
    if model_name in ('RecModel', 'BiasedMF'):
        return model_cls(**common)
    if model_name == 'IPSBiasedMF':
        return model_cls(path=data_loader.path, dataset=data_loader.dataset, M=args.M, **common)
    if model_name == 'DCCF':
        return model_cls(path=data_loader.path, dataset=data_loader.dataset, sentence_model=args.sentence_model,
                         sample_num=args.sample_num, attribute_num=args.attribute_num, std=args.std,
                         n_layers=args.n_layers, **common)
    return None


def main(argv=None):
    init_parser = argparse.ArgumentParser(description='Model')
    init_parser.add_argument('--rank', type=int, default=1, help='1=ranking, 0=rating/click')
    init_parser.add_argument('--data_loader', type=str, default='DataLoader', help='Choose data_loader')
    init_parser.add_argument('--model_name', type=str, default='BaseModel', help='Choose model to run.')
    init_parser.add_argument('--runner', type=str, default='BaseRunner', help='Choose runner')
    init_parser.add_argument('--data_processor', type=str, default='DataProcessor', help='Choose runner')
    init_args, _ = init_parser.parse_known_args(argv)
    data_loader_cls = CLASSES[init_args.data_loader]
    if init_args.model_name not in CLASSES:
        logging.error('Unknown Model: ' + init_args.model_name)
        return
    model_cls = CLASSES[init_args.model_name]
    init_args.runner_name = 'BaseRunner'
    runner_cls = CLASSES[init_args.runner_name]
    dp_cls = CLASSES[init_args.data_processor]

    parser = argparse.ArgumentParser(description='')
    parser = utils.parse_global_args(parser)
    parser = data_loader_cls.parse_data_args(parser)
    parser = model_cls.parse_model_args(parser, model_name=init_args.model_name)
    parser = runner_cls.parse_runner_args(parser)
    parser = dp_cls.parse_dp_args(parser)
    args, _ = parser.parse_known_args(argv)

    name = [str(init_args.rank), init_args.model_name, args.dataset, str(args.random_seed),
            'embdim' + str(getattr(args, 'i_vector_size', 0)), 'optimizer=' + args.optimizer, 'epoch=' + str(args.epoch),
            'lr=' + str(args.lr), 'l2=' + str(args.l2), 'dropout=' + str(args.dropout),
            'batch_size=' + str(args.batch_size), 'test_num=' + str(args.test_neg_n)]
    if init_args.model_name == 'IPSBiasedMF':
        name.append('M' + str(args.M))
    if init_args.model_name == 'DCCF':
        name += ['samnum' + str(args.sample_num), 'feanum' + str(args.attribute_num), 'std' + str(args.std)]
    name = '__'.join(name).replace(' ', '__')
    if args.log_file == '../log/log.txt':
        args.log_file = '../log/%s/%s/%s.txt' % (init_args.model_name, args.dataset, name)
    utils.check_dir_and_mkdir(args.log_file)
    if args.result_file == '../result/result.npy':
        args.result_file = '../result/%s.npy' % name
    utils.check_dir_and_mkdir(args.result_file)     # the reference forgets this one
    if args.model_path == '../model/%s/%s.pt' % (init_args.model_name, init_args.model_name):
        args.model_path = '../model/%s/%s.pt' % (init_args.model_name, name)
    utils.check_dir_and_mkdir(args.model_path)

    # several GPUs (python -m torch.distributed.run --nproc-per-node G -m dccf_amd.main ...; new capability, src/main.py:106 is
    # single-GPU): one process per GPU, identical replicas, rank 0 owns the log, the checkpoint, rank.csv and the result file
    rank, world = utils.init_distributed()
    for h in logging.root.handlers[:]:
        logging.root.removeHandler(h)
    if rank == 0:
        logging.basicConfig(filename=args.log_file, level=args.verbose)
        logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
    else:
        logging.basicConfig(filename=os.devnull, level=logging.WARNING)
    if world > 1:
        logging.info('# ranks: %d (one process per GPU, %s data-parallel training)' % (world, getattr(args, 'mp', 'replicated')))
    logging.info(vars(init_args))
    logging.info(vars(args))
    for what, v in (('DataLoader', init_args.data_loader), ('Model', init_args.model_name),
                    ('Runner', init_args.runner_name), ('DataProcessor', init_args.data_processor)):
        logging.info('%s: %s' % (what, v))

    torch.manual_seed(args.random_seed)
    np.random.seed(args.random_seed)
    if world == 1 and 'HIP_VISIBLE_DEVICES' not in os.environ and 'CUDA_VISIBLE_DEVICES' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu     # src/main.py:106
    logging.info('# cuda devices: %d' % torch.cuda.device_count())

    if world > 1 and rank != 0:
        utils.barrier()             # rank 0 loads first: it writes the info / history cache files the others then find
    data_loader = data_loader_cls(path=args.path, dataset=args.dataset, label=args.label, sep=args.sep)
    if world > 1 and rank == 0:
        utils.barrier()
    data_loader.feature_info(include_id=model_cls.include_id, include_item_features=model_cls.include_item_features,
                             include_user_features=model_cls.include_user_features)
    model = build_model(init_args.model_name, model_cls, args, data_loader)
    if model is None:
        logging.error('Unknown Model: ' + init_args.model_name)
        return
    model.apply(model.init_paras)
    model = model.cuda()
    if hasattr(model, 'eval_noise'):
        model.eval_noise = args.eval_noise
    if init_args.rank == 1:
        data_loader.drop_neg()
    data_processor = dp_cls(data_loader, model, rank=init_args.rank, test_neg_n=args.test_neg_n, seed=args.random_seed,
                            fused_eval=bool(args.fused_sampling))
    runner = runner_cls(optimizer=args.optimizer, learning_rate=args.lr, epoch=args.epoch, batch_size=args.batch_size,
                        eval_batch_size=args.eval_batch_size, dropout=args.dropout, l2=args.l2, metrics=args.metric,
                        check_epoch=args.check_epoch, early_stop=args.early_stop, fused_sampling=args.fused_sampling,
                        use_graph=args.use_graph, device_eval=args.device_eval, overlap_opt=args.overlap_opt, mp=args.mp)
    logging.info('Test Before Training = ' + utils.format_metric(
        runner.evaluate(model, data_processor.get_test_data(), data_processor)) + ' ' + ','.join(runner.metrics))
    if args.load > 0:
        model.load_model()
    if args.train > 0:
        runner.train(model, data_processor, skip_eval=args.skip_eval)
    logging.info('Test After Training = ' + utils.format_metric(
        runner.evaluate(model, data_processor.get_test_data(), data_processor, write_rank=True)) + ' ' + ','.join(runner.metrics))
    if runner.device_eval:
        result = runner.predict_device(model, data_processor.get_test_data(), data_processor).cpu().numpy()
    else:
        result = runner.predict(model, data_processor.get_test_data(), data_processor)
    if rank == 0:
        np.save(args.result_file, result)
        logging.info('Save Test Results to ' + args.result_file)
    utils.barrier()
    runner.model, runner.data_processor = model, data_processor      # for callers that keep working with the trained model
    return runner


if __name__ == '__main__':
    main()
