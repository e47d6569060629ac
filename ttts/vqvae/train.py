"""ttts/vqvae/train.py -> ttts_amd.vqvae.train: `main()`, `run(rank, n_gpus, hps)`,
`train_and_evaluate(rank, epoch, hps, nets, optims, schedulers, scaler, loaders, logger, writers, aug)` (reference :44,119,298).
`torchrun --nproc-per-node N -m ttts.vqvae.train [config.json]` is the reference's `python ttts/vqvae/train.py`."""
from ttts_amd.vqvae.train import (SyntheticVqvaeBatches, VqvaeStep, VqvaeTrainer, augment, build_parts, get_hparams,  # noqa: F401
                                  latest_checkpoint_path, load_checkpoint, main, run, sample_like, save_checkpoint,
                                  train_and_evaluate)

if __name__ == "__main__":
    main()
