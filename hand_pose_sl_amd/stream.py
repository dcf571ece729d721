"""Sequence-sharded streams over the GPUs of one node (one process per GPU).

The reference is single-device; this is the MI355X-native
scale-out of its batched caller (infer_utterance_h5.py:97-115 ->
steps/traintest.py:353-391): utterances are independent, so the stream is
partitioned by SEQUENCE (frames inside one sequence are coupled by the 17-frame
receptive field) and every rank runs the fused kernel on its own shard with no
data-path collective.  The only exchange is the optional hand-back of the
(B_local, T, 21, 2) keypoints to rank 0, done as direct peer->root transfers
(`torch.distributed` P2P = RCCL send/recv on ROCm): on a fully connected xGMI
node each peer then uses its own link into rank 0 instead of a ring.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_seq, rank, world):
    """Contiguous block partition: rank r owns [lo, hi).  Sizes differ by at most 1
    and concatenating the shards in rank order restores the stream order."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(n_seq), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_seq, world):
    return [shard_bounds(n_seq, r, world)[1] - shard_bounds(n_seq, r, world)[0] for r in range(world)]


def gather_to_root(y_local, n_seq, group=None, root=0):
    """Hand every rank's keypoints (B_local, T, 21, 2) back to `root`.

    Returns the (n_seq, T, 21, 2) tensor on root (stream order), None elsewhere.
    Shards may be unequal or empty.  Root posts one receive per peer straight
    into its slice of the result, peers post one send: 7 concurrent xGMI links
    into rank 0 on an 8-GPU node, no ring, no padding."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return y_local
    sizes = shard_sizes(n_seq, world)
    if y_local.shape[0] != sizes[rank]:
        raise RuntimeError(f"rank {rank}: shard has {y_local.shape[0]} sequences, expected {sizes[rank]}")
    y_local = y_local.contiguous()
    ops = []
    out = None
    if rank == root:
        out = torch.empty((n_seq,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        lo, hi = shard_bounds(n_seq, rank, world)
        out[lo:hi].copy_(y_local)
        for r in range(world):
            if r == root or sizes[r] == 0:
                continue
            lo, hi = shard_bounds(n_seq, r, world)
            peer = dist.get_global_rank(group, r) if group is not None else r
            ops.append(dist.P2POp(dist.irecv, out[lo:hi], peer, group))
    elif sizes[rank] > 0:
        peer = dist.get_global_rank(group, root) if group is not None else root
        ops.append(dist.P2POp(dist.isend, y_local, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


class ShardedStream:
    """Run a model over this rank's shard of a stream of sequences.

    `model` is any callable (B,T,12,2)->(B,T,21,2) on this rank's device -- in the
    product a `hand_pose_sl_amd.ConvModel`.  `max_batch` bounds one launch."""

    def __init__(self, model, group=None, max_batch=65536):
        self.model = model
        self.group = group
        self.max_batch = int(max_batch)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def local_slice(self, n_seq):
        return shard_bounds(n_seq, self.rank, self.world)

    @torch.no_grad()
    def run_local(self, x_local):
        if x_local.shape[0] <= self.max_batch:
            return self.model(x_local)
        outs = [self.model(x_local[i:i + self.max_batch]) for i in range(0, x_local.shape[0], self.max_batch)]
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def run_pipelined(self, x_local, n_seq, chunk, root=0):
        """Like run(gather=True), but the shard is processed in `chunk`-sequence pieces and each
        piece is handed to root as soon as it is computed: the transfer of piece k (RCCL runs it
        on its own stream) overlaps the kernel of piece k+1.  Returns the full (n_seq, T, 21, 2)
        result on root, None elsewhere; bit-identical to run().

        Every rank walks the same piece index k.  Per piece a peer posts ONE grouped send and root
        ONE grouped batch of receives, one per peer that still has a piece k
        (`dist.batch_isend_irecv` = ncclGroupStart/End around the send/recv calls), so RCCL sees
        matched groups in the same order on both sides of every pair."""
        if self.world == 1:
            return self.run_local(x_local)
        sizes = shard_sizes(n_seq, self.world)
        if x_local.shape[0] != sizes[self.rank]:
            raise RuntimeError(f"rank {self.rank}: shard has {x_local.shape[0]} sequences, expected {sizes[self.rank]}")
        chunk = max(1, int(chunk))

        def peer(r):
            return dist.get_global_rank(self.group, r) if self.group is not None else r

        reqs, keep, out = [], [], None
        bounds = [shard_bounds(n_seq, r, self.world) for r in range(self.world)]
        lo, hi = bounds[self.rank]
        if self.rank == root:
            T = x_local.shape[1]
            out = torch.empty((n_seq, T, 21, 2), dtype=torch.float32, device=x_local.device)
        for a in range(0, max(sizes), chunk):
            ops = []
            if a < hi - lo:
                y = self.model(x_local[a:a + chunk]).contiguous()
                if self.rank == root:
                    out[lo + a:lo + a + y.shape[0]].copy_(y)
                else:
                    keep.append(y)                 # keep the buffer alive until its send completes
                    ops.append(dist.P2POp(dist.isend, y, peer(root), self.group))
            if self.rank == root:
                for r in range(self.world):
                    if r != root and a < sizes[r]:
                        rlo, rhi = bounds[r]
                        ops.append(dist.P2POp(dist.irecv, out[rlo + a:min(rlo + a + chunk, rhi)], peer(r), self.group))
            if ops:
                reqs.extend(dist.batch_isend_irecv(ops))
        for r in reqs:
            r.wait()
        return out

    @torch.no_grad()
    def run(self, x_local, n_seq, gather=True, root=0):
        """x_local = this rank's sequences [lo,hi) of the stream.  Returns the full
        result on root when gather=True (None on other ranks), else the local shard."""
        y = self.run_local(x_local)
        if not gather:
            return y
        return gather_to_root(y, n_seq, self.group, root)



class HostPipeline:
    """Host tensor in, host tensor out, with the PCIe copies hidden behind each other and the
    kernel: the stream is cut into pieces of `chunk` sequences; piece k's host->device copy, piece
    k-1's kernel and piece k-2's device->host copy run concurrently on three HIP streams over
    `depth` device buffer pairs.  The reference's loops move each batch synchronously
    (steps/traintest.py:354-358, :267-273); this is the MI355X-side replacement for callers whose
    data lives in host memory.  Throughput is bounded by the device->host direction
    (168 B/frame over PCIe), not by the kernel.

    `model`: a hand_pose_sl_amd.ConvModel on a CUDA device.  Results are bit-identical to
    `model(x.cuda()).cpu()`."""

    def __init__(self, model, chunk=16384, depth=2):
        if depth < 2:
            raise ValueError("depth must be >= 2")
        self.model, self.chunk, self.depth = model, int(chunk), int(depth)
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise RuntimeError("HostPipeline needs the model on a CUDA device")
        self.s_in, self.s_run, self.s_out = (torch.cuda.Stream(self.dev) for _ in range(3))
        self._bufs = None

    def _buffers(self, T):
        if self._bufs is None or self._bufs[0][0].shape[1] != T:
            self._bufs = [(torch.empty((self.chunk, T, 12, 2), dtype=torch.float32, device=self.dev),
                           torch.empty((self.chunk, T, 21, 2), dtype=torch.float32, device=self.dev))
                          for _ in range(self.depth)]
        return self._bufs

    @torch.no_grad()
    def run(self, x_host, out=None):
        """x_host: (N, T, 12, 2) float32 CPU tensor (pinned memory gives asynchronous copies; pageable
        memory still works, the runtime then stages it).  Returns (N, T, 21, 2) float32 in pinned
        host memory (or fills `out`)."""
        if x_host.device.type != "cpu" or x_host.dim() != 4 or x_host.shape[2:] != (12, 2):
            raise RuntimeError(f"expected a CPU tensor of shape (N, T, 12, 2), got {tuple(x_host.shape)} on {x_host.device}")
        x_host = x_host.to(torch.float32).contiguous()
        N, T = x_host.shape[0], x_host.shape[1]
        y_host = out if out is not None else torch.empty((N, T, 21, 2), dtype=torch.float32, pin_memory=True)
        bufs = self._buffers(T)
        done_run = [None] * self.depth   # kernel of the piece that last used the slot
        done_out = [None] * self.depth   # device->host copy of that piece
        caller = torch.cuda.current_stream(self.dev)
        for s in (self.s_in, self.s_run, self.s_out):
            s.wait_stream(caller)
        for k, a in enumerate(range(0, N, self.chunk)):
            n = min(self.chunk, N - a)
            slot = k % self.depth
            xd, yd = bufs[slot]
            with torch.cuda.stream(self.s_in):
                if done_run[slot] is not None:
                    self.s_in.wait_event(done_run[slot])          # the slot's input is free again
                xd[:n].copy_(x_host[a:a + n], non_blocking=True)
                e_in = torch.cuda.Event(); e_in.record(self.s_in)
            with torch.cuda.stream(self.s_run):
                self.s_run.wait_event(e_in)
                if done_out[slot] is not None:
                    self.s_run.wait_event(done_out[slot])         # the slot's output has left the device
                self.model.forward_into(xd[:n], yd[:n])
                done_run[slot] = torch.cuda.Event(); done_run[slot].record(self.s_run)
            with torch.cuda.stream(self.s_out):
                self.s_out.wait_event(done_run[slot])
                y_host[a:a + n].copy_(yd[:n], non_blocking=True)
                done_out[slot] = torch.cuda.Event(); done_out[slot].record(self.s_out)
        self.s_out.synchronize()
        self.s_run.synchronize()
        return y_host
