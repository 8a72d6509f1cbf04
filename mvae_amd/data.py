"""Datasets for the MLP path: device-resident image sets with the reference's dynamic binarisation semantics
(mt/data/image_reconstruction.py:37-82) and its VaeDataset interface (mt/data/vae_dataset.py:22-44).

torchvision is not available on either box, so MNIST is read straight from the IDX files when they exist under
`<data>/MNIST/raw/`; otherwise a synthetic set of the same shape (mvae_amd.synthetic.digits_like_batches) is used
and the fact is printed.  The whole set lives in HBM (60000x784 uint8 = 47 MB); a batch is a device-side gather plus
`x > U(0,1)` -- no worker processes, no H2D copy per step.
"""
import gzip
import os
import struct
from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import functional as Fn
from . import synthetic


class VaeDataset:

    def __init__(self, batch_size: int, in_dim: int, img_dims: Optional[Tuple[int, ...]]) -> None:
        self.batch_size = batch_size
        self._in_dim = in_dim
        self._img_dims = img_dims

    def reconstruction_loss(self, x_mb_: Tensor, x_mb: Tensor) -> Tensor:
        raise NotImplementedError

    def create_loaders(self):
        raise NotImplementedError

    @property
    def img_dims(self):
        return self._img_dims

    @property
    def in_dim(self) -> int:
        return self._in_dim

    def metrics(self, x_mb_: Tensor, mode: str = "train") -> Dict[str, float]:
        return {}


class DeviceLoader:
    """Iterates (x_mb, y_mb) over a device-resident uint8/float image matrix; `len(loader.dataset)` = #samples."""

    def __init__(self, images: Tensor, labels: Tensor, batch_size: int, train: bool, binarize: bool,
                 seed: Optional[int] = None) -> None:
        self.images, self.labels = images, labels
        self.batch_size, self.train, self.binarize = batch_size, train, binarize
        self.dataset = range(images.shape[0])
        self._gen = None if seed is None else torch.Generator(device=images.device).manual_seed(seed)

    def __len__(self) -> int:
        return (self.images.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple[Tensor, Tensor]]:
        n = self.images.shape[0]
        dev = self.images.device
        order = torch.randperm(n, device=dev, generator=self._gen) if self.train else torch.arange(n, device=dev)
        for lo in range(0, n, self.batch_size):
            idx = order[lo:lo + self.batch_size]
            x = self.images[idx].to(torch.float32)
            if self.images.dtype == torch.uint8:
                x = x / 255.0
            if self.binarize:  # image_reconstruction.py:44-53
                if self.train:
                    x = (x > torch.rand(x.shape, device=dev, generator=self._gen)).to(torch.float32)
                else:
                    x = (x > 0.5).to(torch.float32)
            yield x, self.labels[idx]


def _read_idx(path: str) -> np.ndarray:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as fh:
        magic, = struct.unpack(">I", fh.read(4))
        ndim = magic & 0xFF
        dims = struct.unpack(">" + "I" * ndim, fh.read(4 * ndim))
        return np.frombuffer(fh.read(), dtype=np.uint8).reshape(dims)


def _find(folder: str, stem: str) -> Optional[str]:
    for sub in ("", "MNIST/raw", "raw"):
        for ext in ("", ".gz"):
            p = os.path.join(folder, sub, stem + ext)
            if os.path.isfile(p):
                return p
    return None


class MnistVaeDataset(VaeDataset):

    def __init__(self, batch_size: int, data_folder: str, device="cuda", synthetic_train: int = 60000,
                 synthetic_test: int = 10000) -> None:
        super().__init__(batch_size, img_dims=(-1, 1, 28, 28), in_dim=784)
        self.data_folder, self.device = data_folder, torch.device(device)
        self._n = (synthetic_train, synthetic_test)

    def _load(self, train: bool) -> Tuple[Tensor, Tensor]:
        stem = "train" if train else "t10k"
        pi, pl = _find(self.data_folder, f"{stem}-images-idx3-ubyte"), _find(self.data_folder, f"{stem}-labels-idx1-ubyte")
        if pi and pl:
            imgs = torch.from_numpy(_read_idx(pi).reshape(-1, 784).copy())
            labels = torch.from_numpy(_read_idx(pl).astype(np.int64))
            return imgs.to(self.device), labels.to(self.device)
        n = self._n[0] if train else self._n[1]
        print(f"MNIST IDX files not found under '{self.data_folder}': using {n} synthetic digits-like images.")
        steps = (n + 127) // 128
        x = synthetic.digits_like_batches(steps, 128, seed=11 if train else 13).reshape(-1, 784)[:n]
        # store as probabilities-in-uint8 so that the loader's dynamic binarisation has something to do
        return (x * 230 + 12).to(torch.uint8).to(self.device), torch.zeros(n, dtype=torch.int64, device=self.device)

    def create_loaders(self, seed: Optional[int] = None):
        tr_x, tr_y = self._load(True)
        te_x, te_y = self._load(False)
        return (DeviceLoader(tr_x, tr_y, self.batch_size, True, True, seed),
                DeviceLoader(te_x, te_y, self.batch_size, False, True, seed))

    def reconstruction_loss(self, x_mb_: Tensor, x_mb: Tensor) -> Tensor:
        """Per-ROW sums of BCE-with-logits (the reference returns per-pixel values that every caller immediately sums
        over the last dim, image_reconstruction.py:81-82 + vae.py:109,131); computed by the HIP kernel."""
        return Fn.bce_rows(x_mb_, x_mb)


class CifarVaeDataset(VaeDataset):
    """image_reconstruction.py:115-143: CIFAR-10 as [N, 3072] channel-major floats in [0,1] (ToTensor + flatten), BCE
    with soft targets, no binarisation.  Reads either layout torchvision's CIFAR10 would have left on disk
    (`cifar-10-batches-py` pickles) or the binary release (`cifar-10-batches-bin/*.bin`: 1 label byte + 3072 pixel
    bytes per record); without files, a synthetic U[0,1) set of the same shape.  The set stays in HBM as uint8
    (50000 x 3072 = 154 MB)."""

    def __init__(self, batch_size: int, data_folder: str, device="cuda", synthetic_train: int = 50000,
                 synthetic_test: int = 10000) -> None:
        super().__init__(batch_size, img_dims=(-1, 3, 32, 32), in_dim=3072)
        self.data_folder, self.device = data_folder, torch.device(device)
        self._n = (synthetic_train, synthetic_test)

    def _read(self, train: bool) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        import pickle
        py = os.path.join(self.data_folder, "cifar-10-batches-py")
        names = [f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]
        if all(os.path.isfile(os.path.join(py, n)) for n in names):
            xs, ys = [], []
            for n in names:
                with open(os.path.join(py, n), "rb") as fh:
                    d = pickle.load(fh, encoding="latin1")
                xs.append(np.asarray(d["data"], dtype=np.uint8).reshape(-1, 3072))
                ys.append(np.asarray(d.get("labels", d.get("fine_labels")), dtype=np.int64))
            return np.concatenate(xs), np.concatenate(ys)
        bn = os.path.join(self.data_folder, "cifar-10-batches-bin")
        names = [f"data_batch_{i}.bin" for i in range(1, 6)] if train else ["test_batch.bin"]
        if all(os.path.isfile(os.path.join(bn, n)) for n in names):
            rec = np.concatenate([np.fromfile(os.path.join(bn, n), dtype=np.uint8) for n in names]).reshape(-1, 3073)
            return rec[:, 1:].copy(), rec[:, 0].astype(np.int64)
        return None

    def _load(self, train: bool) -> Tuple[Tensor, Tensor]:
        got = self._read(train)
        if got is not None:
            return torch.from_numpy(got[0]).to(self.device), torch.from_numpy(got[1]).to(self.device)
        n = self._n[0] if train else self._n[1]
        print(f"CIFAR-10 files not found under '{self.data_folder}': using {n} synthetic U[0,1) images.")
        g = torch.Generator().manual_seed(21 if train else 23)
        x = torch.randint(0, 256, (n, 3072), generator=g, dtype=torch.uint8)
        return x.to(self.device), torch.zeros(n, dtype=torch.int64, device=self.device)

    def create_loaders(self, seed: Optional[int] = None):
        tr_x, tr_y = self._load(True)
        te_x, te_y = self._load(False)
        return (DeviceLoader(tr_x, tr_y, self.batch_size, True, False, seed),
                DeviceLoader(te_x, te_y, self.batch_size, False, False, seed))

    def reconstruction_loss(self, x_mb_: Tensor, x_mb: Tensor) -> Tensor:
        return Fn.bce_rows(x_mb_, x_mb)  # per-row sums, see MnistVaeDataset.reconstruction_loss


def create_dataset(dataset_type: str, *args, **kwargs) -> VaeDataset:  # mt/data/__init__.py:32-42
    if dataset_type == "mnist":
        return MnistVaeDataset(*args, **kwargs)
    if dataset_type == "cifar":
        return CifarVaeDataset(*args, **kwargs)
    if dataset_type in ("bdp", "omniglot"):
        raise NotImplementedError(f"dataset '{dataset_type}' is not part of the MI355X hot-path build yet")
    raise ValueError(f"Unknown dataset type: '{dataset_type}'.")
