"""Checkpoint / file-name helpers of the explainer boundary (host side, not the hot path).

Same naming rules and on-disk formats as the reference so checkpoints and `.npy` outputs interchange:
  gen_prefix / gen_explainer_prefix   utils/io_utils.py:37-60
  create_filename                     utils/io_utils.py:63-78
  load_ckpt                           utils/io_utils.py:106-125 (torch >= 2.6 needs weights_only=False: the
                                      reference pickles the optimizer object and numpy arrays)
"""
import os

import torch


def gen_prefix(args):
    name = args.bmname if getattr(args, "bmname", None) is not None else args.dataset
    name += "_" + args.method
    name += "_h" + str(args.hidden_dim) + "_o" + str(args.output_dim)
    if not args.bias:
        name += "_nobias"
    if len(args.name_suffix) > 0:
        name += "_" + args.name_suffix
    return name


def gen_explainer_prefix(args):
    name = gen_prefix(args) + "_explain"
    if len(args.explainer_suffix) > 0:
        name += "_" + args.explainer_suffix
    return name


def create_filename(save_dir, args, isbest=False, num_epochs=-1):
    filename = os.path.join(save_dir, gen_prefix(args))
    os.makedirs(filename, exist_ok=True)
    if isbest:
        filename = os.path.join(filename, "best")
    elif num_epochs > 0:
        filename = os.path.join(filename, str(num_epochs))
    return filename + ".pth.tar"


def load_ckpt(args, isbest=False):
    filename = create_filename(args.ckptdir, args, isbest)
    if not os.path.isfile(filename):
        print("Checkpoint does not exist: {}".format(filename))
        print("Train one with the reference:  python train.py --dataset=DATASET_NAME")
        raise Exception("File not found.")          # same error behaviour as the reference (io_utils.py:124)
    return torch.load(filename, map_location="cpu", weights_only=False)
