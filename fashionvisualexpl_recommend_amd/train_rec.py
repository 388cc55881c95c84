"""CLI mirror of the reference's src/train_rec.py:17-93 for the two in-scope models.

Same flag names and defaults for every flag BPRMF/VBPR consume; new flags: --optimizer, --dtype, --init_seed.
Run as `python -m fashionvisualexpl_recommend_amd.train_rec --rec bprmf --dataset <name> ...`.
"""
import argparse
import os

from . import configs


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
        if args.rec == 'bprmf':
            model = BPRMF(data, args)
        elif args.rec == 'vbpr':
            model = VBPR(data, args)
        else:
            raise NotImplementedError('Not implemented or unknown Recommender Model.')        # train_rec.py:86
        out.append(model.train())
        print('END REGULARIZATION')
        print('--------------------------------------------------------------------')
    return out


if __name__ == '__main__':
    train()
