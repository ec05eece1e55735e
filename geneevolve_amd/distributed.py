"""Cross-GPU migration: one process per GPU, one population per process, variable-count
all-to-all of complete individual records over torch.distributed (backend "nccl" = RCCL over
xGMI on a multi-GPU node; "gloo" for CPU rehearsal).

Reference semantics (Simulation::ras_do_migration, reference src/Simulation.cpp:877-989): the
HOST decides who moves; migrants are erased from their origin (stayers keep their order,
:960-966) and appended to their destination in (origin ascending, sample order) order (:971-981).
This module only moves the rows: `outgoing[j]` = positions (in the reference's order) of the
individuals of THIS rank's population that move to population j.
"""
import numpy as np
import torch
import torch.distributed as dist


def _fence(on_gpu, trace):
    """Order the HIP library behind torch's stream.  The library launches on its own hipStreamNonBlocking streams, which take
    no implicit order from torch's current stream; an NCCL/RCCL collective only ENQUEUES on torch's stream and returns to the
    host.  So before the library reads a buffer a collective (or any torch op) wrote, and before a collective sends a
    buffer the library wrote (its export entry points already wait for their own stream), the host waits for torch's
    current stream.  One host wait per exchange; the payload of a migration step is a few hundred MB, the wait is its
    transfer time, which the step needs anyway."""
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    if trace is not None:
        trace.append("fence"); trace.append(0.0)


def migrate_all_to_all(ctx, outgoing, local_pop, device=None, group=None, trace=None):
    """ctx: GevContext whose population `local_pop` lives on this rank.  outgoing: list (len = world)
    of uint64 position arrays (entry [rank] must be empty).  Collective: every rank must call it.
    trace: optional list that receives the order of the steps (tests check the fences around the collectives)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    on_gpu = device is not None and torch.device(device).type == "cuda"
    import time
    t_last = [time.perf_counter()]

    def ev(name):                                        # step name (+ host seconds since the previous step, for the bench's breakdown)
        if trace is not None:
            now = time.perf_counter()
            trace.append(name); trace.append(now - t_last[0]); t_last[0] = now
    outgoing = [np.ascontiguousarray(o, dtype=np.uint64) for o in outgoing]
    assert len(outgoing) == world and len(outgoing[rank]) == 0
    # 1. sizes and counts, exchanged first (variable-length records)
    send_bytes = [ctx.export_size(local_pop, o) if len(o) else 0 for o in outgoing]
    meta_dev = torch.device(device) if (on_gpu and backend == "nccl") else torch.device("cpu")
    meta_out = torch.tensor([[send_bytes[j], len(outgoing[j])] for j in range(world)], dtype=torch.int64, device=meta_dev)
    meta_in = torch.empty_like(meta_out)
    dist.all_to_all_single(meta_in, meta_out, group=group)
    ev("a2a_meta")
    meta_in = meta_in.cpu().numpy()                      # (.cpu() waits for the collective on torch's stream)
    recv_bytes = [int(x) for x in meta_in[:, 0]]; recv_n = [int(x) for x in meta_in[:, 1]]
    # 2. pack every destination's records into one send buffer (device memory for the HIP library)
    buf_dev = torch.device(device) if on_gpu else torch.device("cpu")
    send = torch.empty(max(sum(send_bytes), 16), dtype=torch.uint8, device=buf_dev)
    recv = torch.empty(max(sum(recv_bytes), 16), dtype=torch.uint8, device=buf_dev)
    _fence(on_gpu, trace)                                # the caching allocator may hand out memory with work pending on torch's stream
    off = 0
    for j in range(world):
        if send_bytes[j]:
            ctx.export_rows(local_pop, outgoing[j], send.data_ptr() + off, send_bytes[j])   # returns after its own stream has finished
            ev("export")
        off += send_bytes[j]
    # 3. the exchange; gloo cannot move device memory, so a CPU rehearsal of the GPU path stages through the host
    if on_gpu and backend != "nccl":
        send_x, recv_x = send.cpu(), torch.empty(recv.shape, dtype=torch.uint8)
    else:
        send_x, recv_x = send, recv
    if sum(send_bytes) or sum(recv_bytes):
        dist.all_to_all_single(recv_x[:sum(recv_bytes)] if sum(recv_bytes) else recv_x[:0],
                               send_x[:sum(send_bytes)] if sum(send_bytes) else send_x[:0],
                               output_split_sizes=recv_bytes, input_split_sizes=send_bytes, group=group)
        ev("a2a_payload")
    if recv_x is not recv:
        recv.copy_(recv_x)
    # RCCL has only enqueued the exchange on torch's stream: the library must not read `recv` (nor may `send` be released)
    # before it has completed
    _fence(on_gpu, trace)
    # 4. erase the emigrants, then append immigrants origin by origin (ascending)
    gone = np.concatenate([o for o in outgoing]) if sum(len(o) for o in outgoing) else np.empty(0, dtype=np.uint64)
    if len(gone):
        ctx.remove_rows(local_pop, gone)
        ev("remove")
    off = 0
    for i in range(world):
        if recv_n[i]:
            ctx.import_rows(local_pop, recv.data_ptr() + off, recv_bytes[i], recv_n[i])
            ev("import")
        off += recv_bytes[i]
    del send, recv                                       # import_rows returned after its copies finished: safe to release
    return recv_n


class _DevicePtr:
    """a device pointer of the library as a zero-copy tensor source (__cuda_array_interface__)"""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3, "strides": None}


def _device_tensor_f64(ptr, n, device):
    return torch.as_tensor(_DevicePtr(ptr, n), device=torch.device(device))


def compute_ad_locus_split(ctx, pop, group=None, device=None):
    """Simulation::ras_compute_AD (reference src/Simulation.cpp:2624-2749) for a population whose chromosomes are split
    over the ranks of `group` (gev_set_chr_active): every rank holds exact zeros for the chromosomes it does not own, so
    an all-reduce(SUM) of the per-chromosome arrays reproduces each entry bit for bit (x + 0 + ... + 0), and the totals are
    re-summed over chromosomes in the reference's order (:2729-2746).  This is the only collective of a locus-split
    population: N x nchr x nphen doubles per generation.
    Returns (additive, dominance, additive_chr, dominance_chr), identical on every rank of the group."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    if backend == "nccl" and device is not None and torch.device(device).type == "cuda" and ctx.L.exports("compute_ad_device"):
        # RCCL: all-reduce the library's own device arrays in place -- no host round trip of the N x nchr x nphen doubles
        pa, pd, n = ctx.compute_ad_device(pop)                      # (returns after the library's stream has finished)
        ta, td = _device_tensor_f64(pa, n, device), _device_tensor_f64(pd, n, device)
        dist.all_reduce(ta, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(td, op=dist.ReduceOp.SUM, group=group)
        torch.cuda.current_stream().synchronize()                   # RCCL only enqueued on torch's stream: the library reads the arrays next
        return ctx.ad_finish_device(pop)                            # chromosome-ordered FP64 sums on the device (:2729-2746)
    _, _, addc, domc = ctx.compute_ad(pop, per_chr=True)
    both = np.stack([addc, domc])                                   # [2][n][nchr][nphen]
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    t = torch.from_numpy(both).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    both = t.cpu().numpy()
    addc, domc = both[0], both[1]
    add = np.zeros(addc.shape[:1] + addc.shape[2:]); dom = np.zeros_like(add)
    for k in range(addc.shape[1]):                                  # sequential FP64 sums, chromosome order
        add = add + addc[:, k, :]; dom = dom + domc[:, k, :]
    return add, dom, addc, domc
