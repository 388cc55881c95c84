"""CLI mirror of the reference's src/train_rec.py:17-93 for the two in-scope models.

Same flag names and defaults for every flag BPRMF/VBPR consume; new flags: --optimizer, --dtype, --init_seed.
Run as `python -m fashionvisualexpl_recommend_amd.train_rec --rec bprmf --dataset <name> ...`.
"""
import argparse
import os

from . import configs

_last_model = None


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Run train of the Recommender Model.")
    parser.add_argument('--gpu', type=int, default=0, help='HIP device ordinal (the reference default -1 = CPU has '
                                                            'no counterpart: this engine is GPU-only)')
    parser.add_argument('--best_metric', type=str, default='ndcg')
    parser.add_argument('--dataset', nargs='?', default='amazon_baby', help='dataset name')
    parser.add_argument('--rec', nargs='?', default="vbpr", help="bprmf | vbpr")
    parser.add_argument('--batch_size', type=int, default=256, help='batch_size')
    parser.add_argument('--top_k', type=int, default=20, help='top-k of recommendation.')
    parser.add_argument('--epochs', type=int, default=200, help='Number of epochs.')
    parser.add_argument('--verbose', type=int, default=-1, help='number of epochs to store model parameters.')
    parser.add_argument('--batch_eval', type=int, default=128, help='batch size on items for evaluation.')
    parser.add_argument('--lr', type=float, default=0.001, help='Learning rate.')
    parser.add_argument('--validation', type=bool, default=True, help='True to use validation set, False otherwise')
    parser.add_argument('--restore_epochs', type=int, default=1)
    parser.add_argument('--list_of_regs', nargs='+', type=float, default=[0.0], help='list of regularization terms')
    parser.add_argument('--cnn_model', nargs='?', default='vgg19', help='Model used for feature extraction.')
    parser.add_argument('--output_layer', nargs='?', default='fc2', help='Output layer for feature extraction.')
    parser.add_argument('--embed_k', type=int, default=128, help='Embedding size.')
    parser.add_argument('--embed_d', type=int, default=20, help='size of low dimensionality for visual features')
    parser.add_argument('--reg', type=float, default=0, help='regularization')
    # not in the reference
    parser.add_argument('--optimizer', default='adam_tf23', choices=['adam_tf23', 'sgd'])
    parser.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16', 'fp8'],
                        help='storage type of the feature table F (fp8 = OCP e4m3fn codes of f*448)')
    parser.add_argument('--init_seed', type=int, default=0)
    parser.add_argument('--world_size', type=int, default=1,
                        help='> 1: one process per GPU under torch.distributed.run (reads RANK / WORLD_SIZE / LOCAL_RANK)')
    parser.add_argument('--shard', default='item', choices=['item', 'user'],
                        help='item: VBPR, items + features range-partitioned, users replicated; user: BPRMF, users partitioned')
    parser.add_argument('--sampler', default='ref_stream', choices=['ref_stream', 'philox'],
                        help="ref_stream: the reference's MT19937 index stream (single GPU); philox: device epoch walk "
                             "(always used with --world_size > 1: negatives stay GPU-local)")
    parser.add_argument('--dense_reduce', default='gather', choices=['gather', 'allreduce'],
                        help='multi-GPU VBPR: how the gradient of E / beta-prime is summed over the ranks (dist.py)')
    parser.add_argument('--dist_backend', default='nccl', choices=['nccl', 'gloo'], help='nccl == RCCL on ROCm')
    parser.add_argument('--data_root', default=None, help="overrides the reference's '../data'")
    parser.add_argument('--results_root', default=None, help="overrides the reference's '../results'")
    return parser.parse_args(argv)


def train(argv=None):
    args = parse_args(argv)
    configs.set_roots(args.data_root, args.results_root)
    import torch
    from .dataset import DataLoader
    from .models import BPRMF, VBPR
    os.makedirs(os.path.join(configs.results_dir(), args.dataset, args.rec), exist_ok=True)     # train_rec.py:52-55
    os.makedirs(os.path.join(configs.weight_dir(), args.dataset, args.rec), exist_ok=True)
    world = int(args.world_size)
    if world > 1:                                                                               # one process per GPU
        import torch.distributed as dist
        if int(os.environ.get("WORLD_SIZE", "1")) != world:
            raise SystemExit("--world_size %d needs `python -m torch.distributed.run --nproc-per-node %d ...`" % (world, world))
        args.gpu = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("BPRX_ONE_GPU") != "1" else 0
        torch.cuda.set_device(args.gpu)
        if not dist.is_initialized():
            dist.init_process_group(backend=args.dist_backend)
    torch.cuda.set_device(args.gpu)                                                             # train_rec.py:57
    out = []
    for it, current_reg in enumerate(list(args.list_of_regs)):                                  # train_rec.py:60
        print('--------------------------------------------------------------------')
        print('ITERATION %d/%d WITH REGULARIZATION: %f' % (it + 1, len(list(args.list_of_regs)), current_reg))
        data = DataLoader(params=args)
        print("Training {0} on {1}".format(args.rec, args.dataset))
        print("Parameters:")
        args.reg = current_reg                                                                  # train_rec.py:69
        for arg in vars(args):
            print("\t- " + str(arg) + " = " + str(getattr(args, arg)))
        print("\n")
        if world > 1 and args.rec == 'vbpr' and args.shard == 'item':
            from .sharded import ShardedVBPR
            model = ShardedVBPR(data, args)
        elif world > 1 and args.rec == 'bprmf' and args.shard == 'user':
            from .sharded import ShardedBPRMF
            model = ShardedBPRMF(data, args)
        elif world > 1:
            raise NotImplementedError('--world_size > 1 from this CLI: --rec vbpr --shard item, or --rec bprmf --shard user')
        elif args.rec == 'bprmf':
            model = BPRMF(data, args)
        elif args.rec == 'vbpr':
            model = VBPR(data, args)
        else:
            raise NotImplementedError('Not implemented or unknown Recommender Model.')        # train_rec.py:86
        out.append(model.train())
        global _last_model
        _last_model = model                                        # (tests / interactive use)
        print('END REGULARIZATION')
        print('--------------------------------------------------------------------')
    return out


if __name__ == '__main__':
    train()
