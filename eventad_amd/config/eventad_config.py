"""Mirror of `config.eventad_config` (reference config/eventad_config.py:5-134): same flags and
defaults, same YAML merge rule (a YAML key fills an attribute only when the flag is absent).
The default --config points at the dagr-s-dsec.yaml shipped next to this file, because the
reference tree does not contain the YAML its default names."""
import argparse
from pathlib import Path

import yaml

_HERE = Path(__file__).resolve().parent
S = argparse.SUPPRESS


def parse_config(args, config: Path):
    with Path(config).open() as f:
        for k, v in yaml.load(f, Loader=yaml.SafeLoader).items():
            if k not in args:
                setattr(args, k, v)
    return args


def build_parser():
    p = argparse.ArgumentParser(description="EventAD Model Training")
    p.add_argument("--dataset_directory", type=Path, default="./data/detector/ROL")
    p.add_argument("--output_directory", type=Path, default="./log")
    p.add_argument("--checkpoint", type=Path, default="./checkpoints/detector/dagr_s_50.pth")
    p.add_argument("--img_net", default="resnet50", type=str)
    p.add_argument("--img_net_checkpoint", type=Path, default=S)
    p.add_argument("--config", type=Path, default=_HERE / "dagr-s-dsec.yaml")
    p.add_argument("--use_image", default=True, action="store_true")
    p.add_argument("--no_events", action="store_true")
    p.add_argument("--keep_temporal_ordering", action="store_true")
    p.add_argument("--split", default="./config/rol_split.yaml")
    for name, typ in (("task", str), ("dataset", str), ("radius", float), ("time_window_us", int),
                      ("max_neighbors", int), ("n_nodes", int), ("activation", str), ("edge_attr_dim", int),
                      ("aggr", str), ("kernel_size", int), ("pooling_aggr", str), ("base_width", float),
                      ("after_pool_width", float), ("net_stem_width", float), ("yolo_stem_width", float),
                      ("num_scales", int), ("weight_decay", float), ("clip", float), ("aug_p_flip", float),
                      ("aug_trans", float), ("aug_zoom", float), ("l_r", float), ("tot_num_epochs", int)):
        p.add_argument("--" + name, default=S, type=typ)
    p.add_argument("--pooling_dim_at_output", default=S)
    p.add_argument("--batch_size", default=6, type=int)
    p.add_argument("--optimizer", default="adam", type=str)
    p.add_argument("--lr", default=0.003, type=float)
    p.add_argument("--lr_scheduler", default="cosine", type=str)
    p.add_argument("--epochs", default=100, type=int)
    p.add_argument("--no_eval", action="store_true")
    p.add_argument("--run_test", default=True, action="store_true")
    p.add_argument("--num_interframe_steps", type=int, default=6)
    p.add_argument("--x_dim", type=int, default=64)
    p.add_argument("--h_dim", type=int, default=256)
    p.add_argument("--n_frames", type=int, default=100)
    p.add_argument("--fps", type=float, default=20.0)
    p.add_argument("--experiment_name", type=str, default="eventad_dagr_experiment")
    p.add_argument("--output_dir", type=str, default="./output")
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--min_lr", type=float, default=1e-6)
    p.add_argument("--grad_clip", type=float, default=1.0)
    p.add_argument("--pretrained_model", type=str, default="")
    p.add_argument("--height", type=int, default=720)
    p.add_argument("--width", type=int, default=1080)
    p.add_argument("--eventad_config", type=str, default="")
    return p


def parse_eventad_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.config:
        args = parse_config(args, args.config)
    if args.eventad_config:
        args = parse_config(args, Path(args.eventad_config))
    args.dataset_directory = Path(args.dataset_directory)
    args.output_directory = Path(args.output_directory)
    if getattr(args, "checkpoint", None):
        args.checkpoint = Path(args.checkpoint)
    args.num_workers = 4
    for k, v in (("aug_p_flip", 0.5), ("aug_trans", 0.1), ("aug_zoom", 1.5)):
        if not hasattr(args, k):
            setattr(args, k, v)
    return args
