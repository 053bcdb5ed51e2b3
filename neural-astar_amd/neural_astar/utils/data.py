"""Dataset side of the hot path (SURVEY.md section 8f "next #4"): same public surface as the reference's ``utils/data.py``
(``MazeDataset``, ``create_dataloader``, ``WarCraftDataset``, ``create_warcraft_dataloader``, ``visualize_results``), so that
``scripts/train.py`` / ``create_gif.py`` run unmodified against this package -- plus a device-resident batch path.

Why it is here: ``MazeDataset.__getitem__`` (reference ``utils/data.py:152-220``) recomputes three distance percentiles and
rolls the optimal policy out cell by cell in Python for every sample; once a 100-map search step takes 0.4 ms that loader is
the training bottleneck.  This implementation

* computes the percentile thresholds ONCE per map at load time (they depend on the map only);
* keeps the per-sample API and its random stream: ``__getitem__`` makes the same two ``np.random`` draws in the same order as
  the reference (:215,:217), so a seeded run (``set_global_seeds``) sees the identical sequence of start cells;
* adds ``DeviceMazeBatches``: the whole split lives in HBM, a batch is sampled with three device ops and its optimal
  trajectories come from ONE ``nastar_policy_rollout`` launch (``include/nastar.h``) -- no host work per sample.

File layout (planning-datasets): ``arr_{0,4,8}`` map_designs [N,W,W], ``arr_{1,5,9}`` goal_maps [N,1,W,W], ``arr_{2,6,10}``
opt_policies [N,A,1,W,W] one-hot actions, ``arr_{3,7,11}`` opt_dists [N,1,W,W] (negative distances, minimum on obstacles) for
train / valid / test.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import numpy as np
import torch
import torch.utils.data as data

from ..planner.differentiable_astar import AstarOutput  # noqa: F401  (re-exported: the reference's data.py imports it too)

# action index -> (dy, dx): the dataset's "moore" action order (reference utils/data.py:232-241)
ACTION_MOVES = ((-1, 0), (0, 1), (0, -1), (1, 0), (-1, 1), (-1, -1), (1, 1), (1, -1))
_SPLIT_OFFSET = {"train": 0, "valid": 4, "test": 8}


def _grid(x: torch.Tensor, nrow: int = 8, padding: int = 2) -> torch.Tensor:
    """[B,C,H,W] -> [3,Hg,Wg] image grid with the geometry of torchvision.utils.make_grid's defaults (nrow 8, padding 2, zeros
    between tiles, 1-channel inputs repeated to 3), which the reference's visualize_results relies on (:33-35)."""
    x = x.detach().float().cpu()
    if x.shape[1] == 1:
        x = x.expand(-1, 3, -1, -1)
    B, C, H, W = x.shape
    if B == 1:
        return x[0]
    cols = min(nrow, B)
    rows = (B + cols - 1) // cols
    out = torch.zeros((C, rows * (H + padding) + padding, cols * (W + padding) + padding))
    for k in range(B):
        r, c = divmod(k, cols)
        y0, x0 = r * (H + padding) + padding, c * (W + padding) + padding
        out[:, y0:y0 + H, x0:x0 + W] = x[k]
    return out


def visualize_results(map_designs: torch.Tensor, planner_outputs, scale: int = 1) -> np.ndarray:
    """RGB uint8 picture of a batch: maps in grey, explored cells green, path red (reference utils/data.py:14-50)."""
    if isinstance(planner_outputs, dict):
        histories, paths = planner_outputs["histories"], planner_outputs["paths"]
    else:
        histories, paths = planner_outputs.histories, planner_outputs.paths
    img = _grid(map_designs).permute(1, 2, 0).clone()
    h = _grid(histories).permute(1, 2, 0)
    p = _grid(paths).permute(1, 2, 0)
    img[h[..., 0] == 1] = torch.tensor([0.2, 0.8, 0.0])
    img[p[..., 0] == 1] = torch.tensor([1.0, 0.0, 0.0])
    out = (img.numpy() * 255.0).astype("uint8")
    if scale > 1:
        out = np.repeat(np.repeat(out, scale, axis=0), scale, axis=1)  # nearest-neighbour enlargement
    return out


def start_thresholds(opt_dists: np.ndarray, pcts: np.ndarray) -> np.ndarray:
    """[N,1,W,W] negative distances -> [N, len(pcts)] descending thresholds: ``np.percentile`` of the non-obstacle values at
    ``100 * (1 - pcts)`` (reference :212-214; the map's minimum marks obstacles)."""
    N = opt_dists.shape[0]
    th = np.empty((N, len(pcts)), np.float64)
    flat = opt_dists.reshape(N, -1)
    for n in range(N):
        v = flat[n]
        th[n] = np.percentile(v[v > v.min()], 100.0 * (1.0 - pcts))
    return th


class MazeDataset(data.Dataset):
    """Shortest-path problems from a planning-datasets ``.npz`` (same constructor, attributes and item layout as the reference:
    ``map_design [1,W,W]``, ``start_map [num_starts,W,W]``, ``goal_map [1,W,W]``, ``opt_traj [num_starts,W,W]`` float32)."""

    def __init__(self, filename: str, split: str, pct1: float = 0.55, pct2: float = 0.70, pct3: float = 0.85,
                 num_starts: int = 1):
        assert filename.endswith("npz")
        self.filename = filename
        self.dataset_type = split
        self.pcts = np.array([pct1, pct2, pct3, 1.0])
        self.num_starts = num_starts
        self.map_designs, self.goal_maps, self.opt_policies, self.opt_dists = self._process(filename)
        self.num_actions = self.opt_policies.shape[1]
        self.num_orient = self.opt_policies.shape[2]
        self.thresholds = start_thresholds(self.opt_dists, self.pcts)  # [N,4], computed once

    def _process(self, filename: str):
        with np.load(filename) as f:
            i = _SPLIT_OFFSET[self.dataset_type]
            arrs = [f[f"arr_{i + k}"].astype(np.float32) for k in range(4)]
        name = {"train": "Train", "valid": "Validation", "test": "Test"}[self.dataset_type]
        print(f"Number of {name} Samples: {arrs[0].shape[0]}")
        print(f"\tSize: {arrs[0].shape[1]}x{arrs[0].shape[2]}")
        return arrs

    def __len__(self) -> int:
        return self.map_designs.shape[0]

    def __getitem__(self, index: int):
        goal_map = self.goal_maps[index]
        starts = [self._random_start(index) for _ in range(self.num_starts)]
        start_map = np.zeros((self.num_starts,) + goal_map.shape[1:], np.float32)
        start_map.reshape(self.num_starts, -1)[np.arange(self.num_starts), starts] = 1.0
        opt_traj = np.concatenate([self.get_opt_traj(start_map[k:k + 1], goal_map, self.opt_policies[index])
                                   for k in range(self.num_starts)])
        return self.map_designs[index][np.newaxis], start_map, goal_map, opt_traj

    def _random_start(self, index: int) -> int:
        """One start cell for map ``index``: a uniform band r of the three percentile bands, then a uniform cell of that band.
        The two draws are the reference's (:215 ``randint``, :217 ``choice``) in the same order, hence the same seeded stream."""
        v = self.opt_dists[index].reshape(-1)
        th = self.thresholds[index]
        r = np.random.randint(0, len(th) - 1)
        return int(np.random.choice(np.where((v >= th[r + 1]) & (v <= th[r]))[0]))

    def get_random_start_map(self, opt_dist: np.ndarray) -> np.ndarray:
        """Reference-signature form (:201-220) for callers that pass a distance map directly."""
        v = opt_dist.reshape(-1)
        th = np.percentile(v[v > v.min()], 100.0 * (1.0 - self.pcts))
        r = np.random.randint(0, len(th) - 1)
        idx = np.random.choice(np.where((v >= th[r + 1]) & (v <= th[r]))[0])
        out = np.zeros_like(opt_dist)
        out.reshape(-1)[idx] = 1.0
        return out

    def next_loc(self, current_loc: tuple, one_hot_action: np.ndarray) -> tuple:
        dy, dx = ACTION_MOVES[int(np.argmax(one_hot_action))]
        return (current_loc[0], current_loc[1] + dy, current_loc[2] + dx)

    def get_opt_traj(self, start_map: np.ndarray, goal_map: np.ndarray, opt_policy: np.ndarray) -> np.ndarray:
        """Cells visited when following the optimal policy from the start until the goal (goal excluded), reference :171-199."""
        H, W = start_map.shape[-2:]
        act = opt_policy.reshape(opt_policy.shape[0], -1).argmax(0)          # [H*W] action per cell
        cur = int(np.flatnonzero(start_map)[0])
        goal = int(np.flatnonzero(goal_map)[0])
        traj = np.zeros(H * W, np.float32)
        while cur != goal:
            traj[cur] = 1.0
            dy, dx = ACTION_MOVES[act[cur]]
            cur = (cur // W + dy) * W + (cur % W + dx)
            assert traj[cur] == 0.0, "Revisiting the same position while following the optimal policy"
        return traj.reshape(start_map.shape)

    def to_device(self, device) -> "DeviceMazeBatches":
        return DeviceMazeBatches(self, device)


class DeviceMazeBatches:
    """The split resident in HBM; ``sample(indices)`` / iteration yield collated batches ``(map_designs [B,1,W,W], start_maps
    [B,S,W,W], goal_maps [B,1,W,W], opt_trajs [B,S,W,W])`` entirely on the device -- the tuple a ``DataLoader`` over
    ``MazeDataset`` would hand to ``PlannerModule.training_step``."""

    def __init__(self, ds: MazeDataset, device, batch_size: int = 100, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DeviceMazeBatches needs a HIP device (the trajectory roll-out is a HIP kernel)")
        if ds.num_orient != 1:
            raise NotImplementedError("oriented policies (num_orient > 1)")
        self.device, self.num_starts, self.batch_size, self.shuffle, self.generator = dev, ds.num_starts, batch_size, shuffle, generator
        N, W = len(ds), ds.map_designs.shape[-1]
        self.N, self.H, self.W, self.A = N, ds.map_designs.shape[-2], W, ds.num_actions
        self.map_designs = torch.from_numpy(ds.map_designs).to(dev).unsqueeze(1).contiguous()
        self.goal_maps = torch.from_numpy(ds.goal_maps).to(dev).contiguous()
        self.goal_idx = self.goal_maps.reshape(N, -1).argmax(1).to(torch.int32)
        self.opt_policies = torch.from_numpy(ds.opt_policies[:, :, 0]).to(dev).contiguous()          # [N,A,H,W]
        self.opt_dists = torch.from_numpy(ds.opt_dists).to(dev).reshape(N, -1).contiguous()           # [N,HW]
        self.thresholds = torch.from_numpy(ds.thresholds).to(dev)   # [N,4] float64: compared in double, exactly as numpy does
        self.last_status: Optional[torch.Tensor] = None
        self.emit_placement = True  # tag every batch's start_maps with a longest-first placement (see sample())

    def __len__(self) -> int:
        return (self.N + self.batch_size - 1) // self.batch_size

    def candidate_mask(self, idx: torch.Tensor, band: torch.Tensor) -> torch.Tensor:
        """[B] map indices, [B,S] band numbers (0..2) -> bool [B,S,HW]: the band's candidate start cells (reference :216)."""
        od = self.opt_dists[idx][:, None, :].double()
        th = self.thresholds[idx]
        lo = torch.gather(th, 1, band + 1)[..., None]
        hi = torch.gather(th, 1, band)[..., None]
        return (od >= lo) & (od <= hi)

    def sample(self, idx: torch.Tensor, check: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        from .. import _native
        idx = idx.to(self.device)
        B, S, HW = idx.numel(), self.num_starts, self.H * self.W
        band = torch.randint(0, 3, (B, S), device=self.device, generator=self.generator)
        mask = self.candidate_mask(idx, band)
        start_idx = torch.multinomial(mask.reshape(B * S, HW).float(), 1, generator=self.generator).reshape(B, S)
        start_maps = torch.zeros((B, S, HW), dtype=torch.float32, device=self.device)
        start_maps.scatter_(2, start_idx[..., None], 1.0)
        pol = self.opt_policies[idx].contiguous()
        trajs = torch.empty((B, S, self.H, self.W), dtype=torch.float32, device=self.device)
        status = torch.empty((B * S,), dtype=torch.int32, device=self.device)
        si, gi = start_idx.to(torch.int32).contiguous(), self.goal_idx[idx].contiguous()
        lib = _native.load()
        with torch.cuda.device(self.device):
            rc = lib.nastar_policy_rollout(pol.data_ptr(), si.data_ptr(), gi.data_ptr(), B, S, self.A, self.H, self.W,
                                           trajs.data_ptr(), status.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
        _native.check(rc, "nastar_policy_rollout")
        self.last_status = status
        if check and bool((status != 0).any()):
            raise RuntimeError("optimal policy does not lead to the goal for roll-outs " + str(torch.nonzero(status).flatten().tolist()[:16]))
        start_maps = start_maps.reshape(B, S, self.H, self.W)
        if self.emit_placement:
            # a placement for the batch's FIRST search, from data every sample carries: the optimal distance of its start cell
            # (|opt_dists[start]|, reference :127-134,:200-221) -- the longest searches start first (planner/differentiable_astar.py:
            # resolve_placement reads start_maps.placement_order); one gather + one counting-sort launch at batch assembly
            from .. import ops
            ops.attach_order(start_maps, torch.gather(self.opt_dists[idx], 1, start_idx[:, :1]))
        return self.map_designs[idx], start_maps, self.goal_maps[idx], trajs

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]:
        order = (torch.randperm(self.N, device=self.device, generator=self.generator) if self.shuffle
                 else torch.arange(self.N, device=self.device))
        for b0 in range(0, self.N, self.batch_size):
            yield self.sample(order[b0:b0 + self.batch_size])


def create_dataloader(filename: str, split: str, batch_size: int, num_starts: int = 1, shuffle: bool = False) -> data.DataLoader:
    """Host DataLoader with the reference's signature and batch layout (utils/data.py:53-79)."""
    return data.DataLoader(MazeDataset(filename, split, num_starts=num_starts), batch_size=batch_size, shuffle=shuffle, num_workers=0)


def create_device_loader(filename: str, split: str, batch_size: int, device, num_starts: int = 1, shuffle: bool = False,
                         generator: Optional[torch.Generator] = None) -> DeviceMazeBatches:
    """Device-resident equivalent of ``create_dataloader``: iterate it exactly like the DataLoader."""
    return DeviceMazeBatches(MazeDataset(filename, split, num_starts=num_starts), device, batch_size, shuffle, generator)


class WarCraftDataset(data.Dataset):
    """WarCraft terrain maps [N,96,96,3] uint8 + shortest paths [N,12,12]; start top-left, goal bottom-right (reference :270-295)."""

    def __init__(self, dirname: str, split: str):
        self.map_designs = (np.load(f"{dirname}/{split}_maps.npy").transpose(0, 3, 1, 2) / 255.0).astype(np.float32)
        self.paths = np.load(f"{dirname}/{split}_shortest_paths.npy").astype(np.float32)

    def __getitem__(self, index: int):
        opt_traj = self.paths[index][np.newaxis]
        start_map = np.zeros_like(opt_traj)
        start_map[:, 0, 0] = 1
        goal_map = np.zeros_like(opt_traj)
        goal_map[:, -1, -1] = 1
        return self.map_designs[index], start_map, goal_map, opt_traj

    def __len__(self) -> int:
        return self.map_designs.shape[0]


def create_warcraft_dataloader(dirname: str, split: str, batch_size: int, shuffle: bool = False) -> data.DataLoader:
    return data.DataLoader(WarCraftDataset(dirname, split), batch_size=batch_size, shuffle=shuffle, num_workers=0)
