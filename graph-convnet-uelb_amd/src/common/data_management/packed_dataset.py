"""Pre-packed datasets: one binary file per mode instead of three text files per sample, read through a memory
map, uploaded in compact form from pinned double buffers and expanded on the GPU.

The reference re-parses every sample's GML graph and CSV files with networkx on every epoch and then uploads
six dense arrays per batch (``src/common/data_management/dataset_reader.py:68-169``,
``src/gcn/training/base_strategy.py:78-112``); at GPU step times that dominates.  ``pack_dataset`` parses the
reference's on-disk layout once; ``PackedBatchLoader`` yields, per batch, the dict of device tensors that
``prepare_batch_data`` would have produced (same keys, dtypes and values).

File layout (little endian), arrays 64-byte aligned in this order:
    header   64 bytes: magic "GCNPACK1", uint32 version, N, V, C, 40 reserved bytes
    cap      float32 [N,V,V]   edge capacities (the model consumes them as fp32; packing fails if a value
                               does not survive the conversion)
    edges    uint8   [N,V,V]   adjacency as nx.adjacency_matrix gives it
    comm     int32   [N,C,3]   (source, destination, demand)
    flow     int16   [N,C,V]   visiting order of each node on the commodity's path, 0 = not on the path
    lf       float64 [N]       exact maximum load factor (NaN when exact_solution.csv is absent)
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np
import torch

from ...gcn.hip import lib as L
from .dataset_reader import data_dir_for, read_exact_solutions, read_sample

MAGIC = b"GCNPACK1"
VERSION = 1
HEADER = 64


def _layout(N, V, C):
    """name -> (offset, dtype, shape), and the file size."""
    specs = [("cap", np.float32, (N, V, V)), ("edges", np.uint8, (N, V, V)), ("comm", np.int32, (N, C, 3)),
             ("flow", np.int16, (N, C, V)), ("lf", np.float64, (N,))]
    off, out = HEADER, {}
    for name, dt, shape in specs:
        out[name] = (off, np.dtype(dt), shape)
        off += int(np.prod(shape)) * np.dtype(dt).itemsize
        off = (off + 63) & ~63
    return out, off


def pack_file_for(config, mode):
    return data_dir_for(config, mode) / "packed.gcnpack"


def pack_dataset(config, mode, num_data, out_path=None):
    """Parse samples 0..num_data-1 of the reference's on-disk layout once and write the packed file."""
    data_dir = data_dir_for(config, mode)
    out_path = Path(out_path) if out_path else pack_file_for(config, mode)
    adj0, cap0, comm0, flow0 = read_sample(data_dir, 0)
    V, C = cap0.shape[0], len(comm0)
    if V > 32767:
        raise ValueError("node orders are stored as int16")
    layout, size = _layout(num_data, V, C)
    lfs = read_exact_solutions(data_dir)
    tmp = out_path.with_suffix(".tmp")
    with open(tmp, "wb") as f:
        f.write(MAGIC + struct.pack("<4I", VERSION, num_data, V, C) + b"\0" * (HEADER - 8 - 16))
        f.truncate(size)
    mm = np.memmap(tmp, mode="r+", dtype=np.uint8)
    view = {k: mm[o:o + int(np.prod(sh)) * dt.itemsize].view(dt).reshape(sh) for k, (o, dt, sh) in layout.items()}
    for i in range(num_data):
        adj, cap, comm, flow = (adj0, cap0, comm0, flow0) if i == 0 else read_sample(data_dir, i)
        if cap.shape != (V, V) or len(comm) != C or np.asarray(flow).shape != (C, V):
            raise ValueError(f"sample {i} has another shape than sample 0 ({cap.shape}, {len(comm)} commodities)")
        cap32 = cap.astype(np.float32)
        if not np.array_equal(cap32.astype(np.float64), cap):
            raise ValueError(f"sample {i}: a capacity is not representable in fp32")
        flow = np.asarray(flow)
        if flow.min() < 0 or flow.max() > 32767 or not np.isin(adj, (0, 1)).all():
            raise ValueError(f"sample {i}: node order or adjacency out of range")
        view["cap"][i], view["edges"][i], view["comm"][i], view["flow"][i] = cap32, adj, np.asarray(comm), flow
        view["lf"][i] = lfs[i] if lfs is not None and i < len(lfs) else np.nan
    mm.flush()
    del view, mm
    tmp.replace(out_path)
    return out_path


class PackedDataset:
    """Read-only memory map of a packed file; arrays as attributes cap, edges, comm, flow, lf."""

    def __init__(self, path):
        self.path = Path(path)
        with open(self.path, "rb") as f:
            head = f.read(HEADER)
        if head[:8] != MAGIC:
            raise ValueError(f"{path} is not a packed dataset")
        version, self.N, self.V, self.C = struct.unpack("<4I", head[8:24])
        if version != VERSION:
            raise ValueError(f"{path}: version {version}, expected {VERSION}")
        layout, size = _layout(self.N, self.V, self.C)
        if self.path.stat().st_size < size:
            raise ValueError(f"{path} is truncated")
        mm = np.memmap(self.path, mode="r", dtype=np.uint8)
        for k, (o, dt, sh) in layout.items():
            setattr(self, k, mm[o:o + int(np.prod(sh)) * dt.itemsize].view(dt).reshape(sh))


class _Slot:
    """Pinned host staging, device staging and output tensors of one in-flight batch."""

    def __init__(self, B, V, C, dev):
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()
        self.h_cap, self.h_edges = pin((B, V, V), torch.float32), pin((B, V, V), torch.uint8)
        self.h_comm, self.h_flow = pin((B, C, 3), torch.int32), pin((B, C, V), torch.int16)
        d = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.d_edges, self.d_comm, self.d_flow = d((B, V, V), torch.uint8), d((B, C, 3), torch.int32), d((B, C, V), torch.int16)
        self.succ = d((B, C, V), torch.int16)
        self.out = dict(x_edges=d((B, V, V), torch.int64), x_edges_capacity=d((B, V, V), torch.float32),
                        x_nodes=d((B, V, C), torch.int64), y_edges=d((B, V, V, C), torch.int64),
                        batch_commodities=d((B, C, 3), torch.int64), x_commodities=d((B, C), torch.float32))
        self.ready = torch.cuda.Event()
        self.lf = None
        self.indices = None


class PackedBatchLoader:
    """Iterates a packed dataset in batches of device tensors, one batch ahead of the consumer.

    Yields dicts with the keys of ``prepare_batch_data``: x_edges, x_edges_capacity, x_nodes, y_edges,
    batch_commodities, x_commodities, plus load_factor (numpy, host) and indices.  The tensors of a yielded batch
    are reused two batches later: consume (or clone) them before asking for the batch after the next.
    rank / world: this process takes its contiguous shard of every global batch.
    """

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=None, num_data=None, rank=0, world=1):
        self.ds = dataset if isinstance(dataset, PackedDataset) else PackedDataset(dataset)
        self.num_data = self.ds.N if num_data is None else min(int(num_data), self.ds.N)
        self.batch_size, self.shuffle, self.seed = int(batch_size), shuffle, seed
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.GcnHipError("PackedBatchLoader expands batches on the GPU; there is no CPU path")
        base, rem = divmod(self.batch_size, world)
        self.lo = rank * base + min(rank, rem)
        self.local = base + (1 if rank < rem else 0)
        self.max_iter = self.num_data // self.batch_size
        self.copy_stream = torch.cuda.Stream(self.device)
        self.slots = [_Slot(self.local, self.ds.V, self.ds.C, self.device) for _ in range(2)]
        self._epoch = 0

    def __len__(self):
        return self.max_iter

    def _stage(self, slot, idx):
        """Gather the rows of a batch into pinned memory, upload and expand them on the copy stream."""
        lib = L.load()
        ds, B = self.ds, self.local
        if slot.indices is not None:
            slot.ready.synchronize()          # the previous upload from these pinned buffers has left the host
        np.take(ds.cap, idx, axis=0, out=slot.h_cap.numpy())
        np.take(ds.edges, idx, axis=0, out=slot.h_edges.numpy())
        np.take(ds.comm, idx, axis=0, out=slot.h_comm.numpy())
        np.take(ds.flow, idx, axis=0, out=slot.h_flow.numpy())
        slot.lf, slot.indices = np.asarray(ds.lf[idx]), idx
        cs = self.copy_stream
        cs.wait_stream(torch.cuda.current_stream(self.device))        # the consumer is done with this slot's tensors
        with torch.cuda.stream(cs), torch.cuda.device(self.device):
            slot.out["x_edges_capacity"].copy_(slot.h_cap, non_blocking=True)
            slot.d_edges.copy_(slot.h_edges, non_blocking=True)
            slot.d_comm.copy_(slot.h_comm, non_blocking=True)
            slot.d_flow.copy_(slot.h_flow, non_blocking=True)
            o = slot.out
            rc = lib.gcn_unpack_batch(B, ds.V, ds.C, L.ptr(slot.d_edges), L.ptr(slot.d_comm), L.ptr(slot.d_flow),
                                      L.ptr(o["x_edges"]), L.ptr(o["x_nodes"]), L.ptr(o["y_edges"]),
                                      L.ptr(o["batch_commodities"]), L.ptr(o["x_commodities"]), L.ptr(slot.succ),
                                      ctypes_stream(cs))
            L.check(rc, "gcn_unpack_batch")
            slot.ready.record(cs)

    def __iter__(self):
        rng = np.random.default_rng(None if self.seed is None else self.seed + self._epoch)
        self._epoch += 1
        order = np.arange(self.num_data)
        if self.shuffle:
            rng.shuffle(order)
        pick = lambda it: order[it * self.batch_size + self.lo: it * self.batch_size + self.lo + self.local]
        if self.max_iter == 0 or self.local == 0:
            return
        self._stage(self.slots[0], pick(0))
        for it in range(self.max_iter):
            slot = self.slots[it % 2]
            if it + 1 < self.max_iter:
                self._stage(self.slots[(it + 1) % 2], pick(it + 1))
            torch.cuda.current_stream(self.device).wait_event(slot.ready)
            out = dict(slot.out)
            out["load_factor"], out["indices"], out["raw_batch"] = slot.lf, slot.indices, None
            yield out


def ctypes_stream(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)
