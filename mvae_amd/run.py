"""`python -m mvae_amd.run` -- the reference's `python -m mt.examples.run` (run.py:28-186) on the MI355X path.
Same flags, same header / epoch log lines; `--doubles` defaults to False here (the HIP path computes in float32)."""
import argparse
import datetime
import os
import sys

import torch

from . import utils
from .data import create_dataset
from .models import ConvolutionalVAE, FeedForwardVAE
from .trainer import Trainer


def str2bool(v: str) -> bool:  # mt/utils.py:19-26
    v = v.lower()
    if v == "true":
        return True
    if v == "false":
        return False
    raise argparse.ArgumentTypeError(f"Boolean value expected, got '{v}'.")


def main(argv=None) -> None:
    p = argparse.ArgumentParser(description="M-VAE runner (MI355X).")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--data", type=str, default="./data")
    p.add_argument("--batch_size", type=int, default=100)
    p.add_argument("--learning_rate", type=float, default=1e-3)
    p.add_argument("--epochs", type=int, default=500)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--lookahead", type=int, default=50)
    p.add_argument("--model", type=str, default="h2,s2,e2")
    p.add_argument("--architecture", type=str, default="ff")
    p.add_argument("--universal", type=str2bool, default=False)
    p.add_argument("--dataset", type=str, default="mnist")
    p.add_argument("--h_dim", type=int, default=400)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--show_embeddings", type=int, default=0)
    p.add_argument("--export_embeddings", type=int, default=0)
    p.add_argument("--test_every", type=int, default=0)
    p.add_argument("--train_statistics", type=str2bool, default=False)
    p.add_argument("--scalar_parametrization", type=str2bool, default=False)
    p.add_argument("--fixed_curvature", type=str2bool, default=True)
    p.add_argument("--doubles", type=str2bool, default=False)
    p.add_argument("--beta_start", type=float, default=1.0)
    p.add_argument("--beta_end", type=float, default=1.0)
    p.add_argument("--beta_end_epoch", type=int, default=1)
    p.add_argument("--likelihood_n", type=int, default=500)
    args = p.parse_args(argv)

    if args.seed:
        print("Using pre-set random seed:", args.seed)
        utils.set_seeds(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit("mvae_amd needs a HIP device: the MI355X path has no CPU fallback "
                         "(the reference falls back to cpu here, run.py:91-93).")
    # --doubles True (the reference's default, run.py:77): the latent chain of every component runs in float64 between
    # float32 dense layers, through the autograd operators instead of the fused step (models.ModelVAE.float64_chain)
    # Data-parallel training (new functionality; the reference is single-device): under
    #   python -m torch.distributed.run --nproc-per-node N -m mvae_amd.run ...
    # --batch_size stays the GLOBAL batch: every rank trains on its own 1/N of the training set with batch_size / N
    # rows per step, the flat gradient buffer is all-reduced over RCCL before the replicated optimizer step, the epoch
    # statistics are global sums on every rank (identical early-stopping decisions), rank 0 prints.
    from .distributed import init_from_env
    rank, world, local_rank = init_from_env()
    if world > 1:
        if args.batch_size % world:
            raise SystemExit(f"--batch_size={args.batch_size} is not divisible by the {world} ranks")
        args.batch_size //= world
        args.device = f"cuda:{local_rank}"
        if rank != 0:
            sys.stdout = open(os.devnull, "w")
    device = torch.device(args.device)
    print("Running on:", device, flush=True)

    dataset = create_dataset(args.dataset, args.batch_size, args.data, device=device)
    print("#####")
    cur_time = datetime.datetime.utcnow().isoformat()
    components = utils.parse_components(args.model, args.fixed_curvature)
    model_name = utils.canonical_name(components)
    print(f"VAE Model: {model_name}; Epochs: {args.epochs}; Time: {cur_time}; Fixed curvature: {args.fixed_curvature}; "
          f"Dataset: {args.dataset}")
    print("#####", flush=True)
    if args.doubles:  # refused before anything is created (checkpoint directory, model)
        if world > 1 or args.architecture != "ff":
            raise SystemExit("--doubles=True (float64 latent chain) is built for the single-device ff architecture")
        too_big = [f"{type(c).__name__}({c.true_dim})" for c in components if c.true_dim > 8]
        if too_big:  # mvae_component_forward_f64 / _backward_f64 keep a component's vectors in registers: true dim <= 8
            raise SystemExit("--doubles=True: the float64 latent chain is built for components of true dimension <= 8; "
                             f"this model has {', '.join(too_big)} (run with --doubles=False)")
    chkpt_dir = f"./chkpt/vae-{args.dataset}-{model_name}-{cur_time}" + (f"-rank{rank}" if rank else "")
    os.makedirs(chkpt_dir)
    if args.architecture == "ff":
        model_cls = FeedForwardVAE
    elif args.architecture == "conv":
        model_cls = ConvolutionalVAE
    else:
        raise ValueError(f"Unknown --architecture='{args.architecture}'. Possible options: 'ff', 'conv'.")
    model = model_cls(h_dim=args.h_dim, components=components, dataset=dataset,
                      scalar_parametrization=args.scalar_parametrization).to(device)
    if args.seed:
        model.seed_sampler(args.seed + rank)
    if args.doubles:
        model.float64_chain = True
    if world > 1:
        model.enable_data_parallel()
    trainer = Trainer(model, img_dims=dataset.img_dims, chkpt_dir=chkpt_dir, train_statistics=args.train_statistics,
                      show_embeddings=args.show_embeddings, export_embeddings=args.export_embeddings,
                      test_every=args.test_every)
    optimizer = trainer.build_optimizer(learning_rate=args.learning_rate, fixed_curvature=args.fixed_curvature)
    train_loader, test_loader = dataset.create_loaders(seed=args.seed)
    if world > 1:  # equal shards: every rank takes the same number of steps per epoch
        n = (train_loader.images.shape[0] // world) * world
        train_loader.images = train_loader.images[:n][rank::world].contiguous()
        train_loader.labels = train_loader.labels[:n][rank::world].contiguous()
        train_loader.dataset = range(train_loader.images.shape[0])
    betas = utils.linear_betas(args.beta_start, args.beta_end, end_epoch=args.beta_end_epoch, epochs=args.epochs)
    if args.universal:  # the universal training scheme, run.py:139-175
        # pre-training at K = 0 (every `u` component is Euclidean)
        trainer.train_epochs(optimizer=optimizer, train_data=train_loader, eval_data=test_loader,
                             epochs=args.epochs // 2, betas=betas, likelihood_n=0)
        # choose signs: a third hyperbolic, a third spherical, the rest stays Euclidean
        eps = 1e-5
        cn = len(model.components) // 3
        signs = [-1] * cn + [1] * cn + [0] * (len(model.components) - 2 * cn)
        print("Chosen signs:", signs)
        for i, component in enumerate(model.components):
            component._curvature.data += signs[i] * eps
            component._curvature.requires_grad = False
        # ... continue without learning curvature for 10 epochs
        trainer.train_epochs(optimizer=optimizer, train_data=train_loader, eval_data=test_loader, epochs=10,
                             betas=betas, likelihood_n=0)
        # ... then unfix it
        for component in model.components:
            component._curvature.requires_grad = True
        trainer.train_stopping(optimizer=optimizer, train_data=train_loader, eval_data=test_loader,
                               warmup=args.lookahead + 1, lookahead=args.lookahead, betas=betas,
                               likelihood_n=args.likelihood_n, max_epochs=args.epochs)
    else:
        trainer.train_stopping(optimizer=optimizer, train_data=train_loader, eval_data=test_loader,
                               warmup=args.warmup, lookahead=args.lookahead, betas=betas,
                               likelihood_n=args.likelihood_n, max_epochs=args.epochs)
    print(flush=True)
    print("Done.", flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
