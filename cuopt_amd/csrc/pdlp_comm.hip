// Communicators, collectives and the direct peer transport's set-up of sharded solves (one translation unit: the multi-GPU side
// of the device layer).  The kernels of an attempt are launched by the core (pdlp_device.hip), which calls the collectives below.
#include <limits>

#include "pdlp_ctx.hpp"

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// ---- multi-GPU ----------------------------------------------------------------------------------
int pdlpdev_comm_unique_id(uint8_t id[128])
{
  TRY(rccl::load());
  rccl::unique_id u;
  RCCL_TRY(rccl::GetUniqueId(&u));
  memcpy(id, u.internal, 128);
  return 0;
}
int pdlpdev_softcomm_create(int world, uint8_t id[128])
{
  if (world < 1 || world > 16) return fail(-1, "pdlpdev_softcomm_create: world must be in 1..16");
  softcomm::Comm* c = new softcomm::Comm();
  c->world = world;
  c->bufs.assign(world, nullptr), c->scratch.assign(world, nullptr), c->scratch_size.assign(world, 0);
  memset(id, 0, 128);
  memcpy(id, softcomm::kMagic, 8);
  memcpy(id + 8, &c, sizeof(c));
  return 0;
}
// CUOPT_AMD_SHARD_DATAFLOW = owner (default) | allreduce | rsag : see the `rsag` / `owner` fields of the context
int setup_dataflow(pdlpdev_ctx* ctx)
{
  const char* env = getenv("CUOPT_AMD_SHARD_DATAFLOW");
  const std::string flow = env ? env : "owner";  // default since round 3: nothing but the slices themselves travels
  if (flow == "allreduce") return 0;
  if (flow != "rsag" && flow != "owner") return fail(-1, "CUOPT_AMD_SHARD_DATAFLOW must be allreduce, rsag or owner");
  if (ctx->world > 16) return fail(-1, "the sliced-primal dataflows support up to 16 ranks");
  if (!ctx->soft && (!rccl::ReduceScatter || !rccl::AllGather)) return fail(-3, "RCCL: ncclReduceScatter / ncclAllGather missing");
  const int per = (ctx->n + ctx->world - 1) / ctx->world;
  ctx->slice    = (per + 15) & ~15;
  ctx->rsag     = true;  // both keep the primal side in slices inside the attempt loop
  ctx->owner    = flow == "owner";  // ... the column block arrives with pdlpdev_owner_setup
  TRY(dev_alloc(ctx, &ctx->rs_buf, (size_t)ctx->slice + 8));
  TRY(dev_alloc(ctx, &ctx->rs_scal, 8 + 4 * 16));  // [0..3) this rank's sums, [4..7) the ranks' sums, [8..) landed scalars (p2p)
  return 0;
}
int pdlpdev_comm_init(pdlpdev_ctx* ctx, int rank, int world, const uint8_t id[128])
{
  if (memcmp(id, softcomm::kMagic, 8) == 0) {
    softcomm::Comm* c = nullptr;
    memcpy(&c, id + 8, sizeof(c));
    if (!c || c->world != world) return fail(-1, "soft communicator: world mismatch");
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->refs += 1;
    }
    ctx->soft = c;
    ctx->comm = reinterpret_cast<rccl::comm_t>(c);  // marks sharded mode; never passed to RCCL
    ctx->rank = rank, ctx->world = world;
    return setup_dataflow(ctx);
  }
  TRY(rccl::load());
  HIP_TRY(hipSetDevice(ctx->device));
  // A unique id bootstraps exactly ONE communicator per rank; solvers created later with the same (id, rank) -- bench.py
  // makes two per process -- share it.  The ranks of a single-process sharded solve (cuoptamd_solve_sharded: one host
  // thread per device) each get their own; ncclCommInitRank blocks until every rank has joined, so it runs outside the
  // lock.  Communicators live until process exit.
  std::string key((const char*)id, 128);
  key.append((const char*)&rank, sizeof(rank));
  rccl::comm_t comm = nullptr;
  {
    std::lock_guard<std::mutex> lock(comm_cache::mu);
    auto it = comm_cache::map.find(key);
    if (it != comm_cache::map.end()) comm = it->second.comm, it->second.refs += 1;
  }
  if (!comm) {
    rccl::unique_id u;
    memcpy(u.internal, id, 128);
    RCCL_TRY(rccl::CommInitRank(&comm, world, u, rank));
    std::lock_guard<std::mutex> lock(comm_cache::mu);
    comm_cache::map.emplace(key, comm_cache::Entry{comm, 1});
  }
  ctx->comm = comm, ctx->comm_key = key;
  ctx->rank = rank, ctx->world = world;
  return setup_dataflow(ctx);
}
// A rank of a sharded solve failed: nobody may wait for it.  Aborts every communicator this process created from `id`
// (ncclCommAbort ends the collectives in flight; the in-process communicator wakes its barriers) -- the other ranks' next
// collective returns an error instead of blocking.
int pdlpdev_comm_abort(const uint8_t id[128])
{
  if (memcmp(id, softcomm::kMagic, 8) == 0) {
    softcomm::Comm* c = nullptr;
    memcpy(&c, id + 8, sizeof(c));
    if (c) c->abort();
    return 0;
  }
  std::lock_guard<std::mutex> lock(comm_cache::mu);
  for (auto& kv : comm_cache::map)
    if (kv.first.compare(0, 128, std::string((const char*)id, 128)) == 0 && !kv.second.aborted) {
      kv.second.aborted = true;
      if (rccl::CommAbort) (void)rccl::CommAbort(kv.second.comm);
    }
  return 0;
}
// recv[0..count) = sum over the ranks of send[rank * count ..][0..count)   (ncclReduceScatter)
int reduce_scatter(pdlpdev_ctx* ctx, const double* send, double* recv, size_t count)
{
  if (ctx->soft) {
    softcomm::Comm* c = ctx->soft;
    const int r       = ctx->rank;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // my contribution is complete
    c->bufs[r] = const_cast<double*>(send);
    SOFT_BARRIER(c);
    softcomm::Peers peers;
    for (int q = 0; q < c->world; ++q) peers.p[q] = c->bufs[q] + (size_t)r * count;
    const int g = (int)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 1024));
    softcomm::k_combine<<<g, 256, 0, ctx->stream>>>(peers, c->world, count, 0, recv);  // recv is nobody's input
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    SOFT_BARRIER(c);  // nobody still reads the inputs
    return 0;
  }
  RCCL_TRY(rccl::ReduceScatter(send, recv, count, rccl::kFloat64, rccl::kSum, ctx->comm, ctx->stream));
  return 0;
}
// buf[q * count ..][0..count) = rank q's slice, in place (ncclAllGather with sendbuff = recvbuff + rank * count)
int all_gather(pdlpdev_ctx* ctx, double* buf, size_t count)
{
  if (ctx->soft) {
    softcomm::Comm* c = ctx->soft;
    const int r       = ctx->rank;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    c->bufs[r] = buf;
    SOFT_BARRIER(c);
    for (int q = 0; q < c->world; ++q)
      if (q != r)
        HIP_TRY(hipMemcpyAsync(buf + (size_t)q * count, c->bufs[q] + (size_t)q * count, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    SOFT_BARRIER(c);  // nobody still reads my slice
    return 0;
  }
  RCCL_TRY(rccl::AllGather(buf + (size_t)ctx->rank * count, buf, count, rccl::kFloat64, ctx->comm, ctx->stream));
  return 0;
}
int allreduce(pdlpdev_ctx* ctx, double* buf, size_t count, int op)
{
  if (!ctx->comm) return 0;
  if (ctx->soft) {
    softcomm::Comm* c = ctx->soft;
    const int r       = ctx->rank;
    if (c->scratch_size[r] < count) {
      if (c->scratch[r]) (void)hipFree(c->scratch[r]);
      HIP_TRY(hipMalloc((void**)&c->scratch[r], count * sizeof(double)));
      c->scratch_size[r] = count;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // my contribution is complete
    c->bufs[r] = buf;
    SOFT_BARRIER(c);                                // everybody's contribution is complete and published
    softcomm::Peers peers;
    for (int q = 0; q < c->world; ++q) peers.p[q] = c->bufs[q];
    const int g = (int)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 1024));
    softcomm::k_combine<<<g, 256, 0, ctx->stream>>>(peers, c->world, count, op == rccl::kSum ? 0 : 1, c->scratch[r]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    SOFT_BARRIER(c);                                // nobody still reads the inputs
    HIP_TRY(hipMemcpyAsync(buf, c->scratch[r], count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  RCCL_TRY(rccl::AllReduce(buf, buf, count, rccl::kFloat64, op, ctx->comm, ctx->stream));
  return 0;
}


// ---- halo exchange (pdlp_ctx.hpp Halo) -------------------------------------------------------------------------------------------
// need[q * 4 + {0, 1}]: the columns [lo, hi) of rank q's slice that this rank's rows reference; need[q * 4 + {2, 3}]: the positions
// [lo, hi) of rank q's rows in the gathered dual that this rank's columns reference (empty: lo >= hi).  Every rank learns the whole
// table (one all-gather of 4 * world numbers per rank), so that it knows what to send, and all ranks take the same decision.
int halo_setup(pdlpdev_ctx* ctx, const int32_t* need)
{
  const int W = ctx->world, me = ctx->rank;
  pdlpdev_ctx::Halo& H = ctx->halo;
  H.on = false;
  const size_t per = (size_t)4 * W;
  double* wire = nullptr;
  TRY(dev_alloc(ctx, &wire, per * W + 16));
  std::vector<double> mine(per), all(per * W);
  for (size_t i = 0; i < per; ++i) mine[i] = (double)need[i];
  HIP_TRY(hipMemcpyAsync(wire + (size_t)me * per, mine.data(), per * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TRY(all_gather(ctx, wire, per));
  HIP_TRY(hipMemcpyAsync(all.data(), wire, per * W * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  auto range = [&](int r, int q, int kind, int32_t* off, int32_t* cnt) {  // what rank r needs from rank q
    const int32_t lo = (int32_t)all[(size_t)r * per + (size_t)q * 4 + 2 * kind], hi = (int32_t)all[(size_t)r * per + (size_t)q * 4 + 2 * kind + 1];
    *off = lo, *cnt = r != q && hi > lo ? hi - lo : 0;
  };
  int64_t worst = 0;  // the largest volume any rank receives per attempt (entries)
  for (int r = 0; r < W; ++r) {
    int64_t v = 0;
    for (int q = 0; q < W; ++q)
      for (int kind = 0; kind < 2; ++kind) {
        int32_t off, cnt;
        range(r, q, kind, &off, &cnt);
        v += cnt;
      }
    worst = std::max(worst, v);
  }
  const int64_t full = (int64_t)(W - 1) * ((int64_t)ctx->slice + ctx->ypad);
  H.bytes_allgather  = 8 * full;
  const long long want = cuopt_amd::tune_int("shard_halo", -1);
  bool use = W > 1 && want != 0 && (want == 1 || worst * 4 <= full);
  if (!ctx->soft && !ctx->p2p.on && use && (!rccl::Send || !rccl::Recv || !rccl::GroupStart || !rccl::GroupEnd)) {
    // only an EXPLICIT shard_halo=1 makes the missing entry points an error; the automatic rule falls back to the all-gathers (every
    // rank loads the same library, so every rank takes this branch)
    if (want == 1) return fail(-3, "RCCL: ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd missing (CUOPT_AMD_TUNE=shard_halo=1 asked for the halo exchange)");
    use = false;
  }
  for (int kind = 0; kind < 2; ++kind) {
    H.recv_off[kind].assign(W, 0), H.recv_cnt[kind].assign(W, 0), H.send_off[kind].assign(W, 0), H.send_cnt[kind].assign(W, 0);
    for (int q = 0; q < W; ++q) {
      range(me, q, kind, &H.recv_off[kind][q], &H.recv_cnt[kind][q]);
      range(q, me, kind, &H.send_off[kind][q], &H.send_cnt[kind][q]);
    }
  }
  H.bytes = 0;
  for (int kind = 0; kind < 2; ++kind)
    for (int q = 0; q < W; ++q) H.bytes += 8 * (int64_t)H.recv_cnt[kind][q];
  H.on = use;
  if (use && ctx->p2p.on) TRY(p2p_push_ranges(ctx));  // the peer stores carry the halo too: only what the peer's rows / columns reference
  if (getenv("CUOPT_AMD_TIMING"))
    fprintf(stderr, "[cuopt_amd setup]   rank %d: halo %s: %lld B per attempt against %lld B for the two all-gathers\n", me, use ? "on" : "off", (long long)H.bytes,
            (long long)H.bytes_allgather);
  return 0;
}

// the ranges of `buf` (kind 0: xbar, kind 1: the gathered y') this rank needs arrive from their owners, the ranges others need leave
int halo_exchange(pdlpdev_ctx* ctx, int kind, double* buf)
{
  const pdlpdev_ctx::Halo& H = ctx->halo;
  const int W = ctx->world, me = ctx->rank;
  if (ctx->soft) {
    softcomm::Comm* c = ctx->soft;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // my part is complete
    c->bufs[me] = buf;
    SOFT_BARRIER(c);
    for (int q = 0; q < W; ++q)
      if (H.recv_cnt[kind][q] > 0)  // (the peers' buffers share this one's geometry: the same offsets)
        HIP_TRY(hipMemcpyAsync(buf + H.recv_off[kind][q], c->bufs[q] + H.recv_off[kind][q], (size_t)H.recv_cnt[kind][q] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    SOFT_BARRIER(c);  // nobody still reads my part
    return 0;
  }
  RCCL_TRY(rccl::GroupStart());
  for (int q = 0; q < W; ++q) {
    if (H.send_cnt[kind][q] > 0) RCCL_TRY(rccl::Send(buf + H.send_off[kind][q], (size_t)H.send_cnt[kind][q], rccl::kFloat64, q, ctx->comm, ctx->stream));
    if (H.recv_cnt[kind][q] > 0) RCCL_TRY(rccl::Recv(buf + H.recv_off[kind][q], (size_t)H.recv_cnt[kind][q], rccl::kFloat64, q, ctx->comm, ctx->stream));
  }
  RCCL_TRY(rccl::GroupEnd());
  return 0;
}

// Direct peer transport: allocate this rank's landing block and learn where the other ranks' blocks are.
//   in-process communicator: the ranks are contexts of one process (tests: on ONE device) -> a table in the communicator;
//   RCCL: one 128-byte record per rank {IPC handle, process id, pointer, device} all-gathered through the communicator:
//   same process -> the pointer itself (peer access enabled), another process -> hipIpcOpenMemHandle.
// halo exchange over the direct peer transport: the two vector exchanges' descriptors learn, per destination rank, which entries of
// THIS rank's share it needs (halo_setup's send ranges, buffer coordinates -> coordinates of the producing kernel's loop)
int p2p_push_ranges(pdlpdev_ctx* ctx)
{
  pdlpdev_ctx::P2P& P = ctx->p2p;
  const pdlpdev_ctx::Halo& H = ctx->halo;
  const int64_t origin[2] = {(int64_t)ctx->rank * ctx->slice, (int64_t)ctx->rank * ctx->ypad};
  for (int kind = 0; kind < 2; ++kind)
    for (int q = 0; q < ctx->world; ++q) {
      const int64_t lo = (int64_t)H.send_off[kind][q] - origin[kind];
      P.push_host[kind].lo[q] = H.send_cnt[kind][q] > 0 ? (int)lo : 0;
      P.push_host[kind].hi[q] = H.send_cnt[kind][q] > 0 ? (int)(lo + H.send_cnt[kind][q]) : 0;
    }
  HIP_TRY(hipMemcpyAsync(P.push_dev, P.push_host, sizeof(P.push_host), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);  // (the pull kernels of the attempt graphs change)
  ctx->graphs.clear();
  return 0;
}

int p2p_setup(pdlpdev_ctx* ctx)
{
  pdlpdev_ctx::P2P& P = ctx->p2p;
  const size_t W = (size_t)ctx->world;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  P.off_x = 0;
  P.off_y = align(P.off_x + W * (size_t)ctx->slice * sizeof(double));
  P.off_s = align(P.off_y + W * (size_t)ctx->ypad * sizeof(double));
  P.off_f = align(P.off_s + W * 4 * sizeof(double));
  P.bytes = (P.off_f + p2pdev::kKinds * W * sizeof(unsigned long long) + 4095) & ~(size_t)4095;
  HIP_TRY(hipExtMallocWithFlags((void**)&P.base, P.bytes, hipDeviceMallocFinegrained));
  HIP_TRY(hipMemsetAsync(P.base, 0, P.bytes, ctx->stream));  // (on the context's stream -- a non-blocking one, which the null stream's work is not ordered with)
  TRY(dev_alloc(ctx, &P.epoch, 4));
  TRY(dev_alloc(ctx, &P.fault, 4));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->soft) {
    softcomm::Comm* c = ctx->soft;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      if (c->p2p_base.size() != W) c->p2p_base.assign(W, nullptr);
      c->p2p_base[ctx->rank] = P.base;
    }
    SOFT_BARRIER(c);
    for (size_t q = 0; q < W; ++q) P.peers.base[q] = (char*)c->p2p_base[q];
    SOFT_BARRIER(c);  // everybody has read the table before anybody can overwrite it with the next solver's blocks
  } else {
    struct Rec {
      hipIpcMemHandle_t handle;
      unsigned long long pid, ptr, device;
      char pad[128 - sizeof(hipIpcMemHandle_t) - 24];
    };
    static_assert(sizeof(Rec) == 128, "one record = 16 doubles on the wire");
    std::vector<Rec> recs(W);
    Rec mine;
    memset(&mine, 0, sizeof(mine));
    HIP_TRY(hipIpcGetMemHandle(&mine.handle, P.base));
    mine.pid = (unsigned long long)getpid(), mine.ptr = (unsigned long long)(uintptr_t)P.base, mine.device = (unsigned long long)ctx->device;
    double* wire = nullptr;
    TRY(dev_alloc(ctx, &wire, W * 16));
    HIP_TRY(hipMemcpyAsync(wire + (size_t)ctx->rank * 16, &mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    RCCL_TRY(rccl::AllGather(wire + (size_t)ctx->rank * 16, wire, 16, rccl::kFloat64, ctx->comm, ctx->stream));
    HIP_TRY(hipMemcpyAsync(recs.data(), wire, W * sizeof(Rec), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t q = 0; q < W; ++q) {
      if ((int)q == ctx->rank) {
        P.peers.base[q] = P.base;
      } else if (recs[q].pid == mine.pid) {
        if ((int)recs[q].device != ctx->device) {
          const hipError_t e = hipDeviceEnablePeerAccess((int)recs[q].device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(-2, "hipDeviceEnablePeerAccess(%d): %s", (int)recs[q].device, hipGetErrorString(e));
          (void)hipGetLastError();
        }
        P.peers.base[q] = (char*)(uintptr_t)recs[q].ptr;
      } else {
        void* mapped = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&mapped, recs[q].handle, hipIpcMemLazyEnablePeerAccess));
        P.opened.push_back(mapped);
        P.peers.base[q] = (char*)mapped;
      }
    }
  }
  {
    p2pdev::Push h[p2pdev::kKinds];
    const size_t slot[p2pdev::kKinds] = {P.off_x + (size_t)ctx->rank * ctx->slice * sizeof(double), P.off_y + (size_t)ctx->rank * ctx->ypad * sizeof(double),
                                         P.off_s + (size_t)ctx->rank * 4 * sizeof(double)};
    for (int k = 0; k < p2pdev::kKinds; ++k) {
      h[k] = p2pdev::Push{P.peers, ctx->world, ctx->rank, k, slot[k], P.off_f, P.epoch, {}, {}};
      for (int q = 0; q < 16; ++q) h[k].lo[q] = 0, h[k].hi[q] = std::numeric_limits<int>::max();
      P.push_host[k] = h[k];
    }
    TRY(dev_alloc(ctx, &P.push_dev, p2pdev::kKinds));
    HIP_TRY(hipMemcpyAsync(P.push_dev, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  P.on = true;
  return 0;
}

// ---- cross-PROCESS rehearsal of the direct peer transport on ONE device (round-5 review, item 5b) ------------------------------------
// Multi-rank RCCL cannot be brought up on a one-GPU box (it refuses duplicate devices), so the part of p2p_setup that only a second
// PROCESS exercises -- hipIpcGetMemHandle of a fine-grained landing block, hipIpcOpenMemHandle with lazy peer access in another
// process, the producers' system-scope stores (p2pdev::put), the flag raised with release semantics (p2pdev::raise) and the consumer's
// wait (p2pdev::wait_flags) across the process boundary -- is reachable here without a communicator: tests/test_p2p_transport_gpu.py
// starts a second process, hands it the 64-byte handle through a pipe, and lets it store into this process's block.
//   block = [count doubles | 2 flags (exchange 0, world 2)]
static __global__ void __launch_bounds__(256) k_ipc_store(p2pdev::Push T, int count, double seed)
{
  for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) p2pdev::put(T.slot(0) + i, seed + (double)i);
}
static __global__ void k_ipc_raise(p2pdev::Push T)  // the NEXT kernel of the stream publishes (p2pdev::raise's contract)
{
  p2pdev::raise(&T);
}
static __global__ void __launch_bounds__(256) k_ipc_wait_and_check(const double* __restrict__ land, const unsigned long long* __restrict__ flags,
                                                            const unsigned long long* __restrict__ epoch, int count, double seed, int* __restrict__ bad)
{
  if (!p2pdev::wait_flags(flags, 2, 0, epoch)) {
    if (threadIdx.x == 0) atomicAdd(bad, 1 << 20);  // patience ran out
    return;
  }
  int wrong = 0;
  for (int i = threadIdx.x; i < count; i += 256) wrong += __builtin_nontemporal_load(land + i) != seed + (double)i;
  if (wrong) atomicAdd(bad, wrong);
}
extern "C" {
// the OWNER: a zeroed fine-grained block of count doubles + flags, its IPC handle (64 bytes) and its address
int pdlpdev_debug_ipc_export(int device, int count, uint8_t handle[64], void** base)
{
  HIP_TRY(hipSetDevice(device));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
  const size_t bytes = ((size_t)count * 8 + 2 * 8 * 2 + 4095) & ~(size_t)4095;
  HIP_TRY(hipExtMallocWithFlags(base, bytes, hipDeviceMallocFinegrained));
  HIP_TRY(hipMemset(*base, 0, bytes));
  HIP_TRY(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, *base));
  memcpy(handle, &h, 64);
  return 0;
}
// the OTHER process: opens the block, stores seed + i into entry i with the transport's own put, raises "rank 1"'s flag of exchange 0
int pdlpdev_debug_ipc_store(int device, const uint8_t handle[64], int count, double seed)
{
  HIP_TRY(hipSetDevice(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  void* mapped = nullptr;
  HIP_TRY(hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess));
  unsigned long long* epoch = nullptr;
  HIP_TRY(hipMalloc((void**)&epoch, 4 * sizeof(unsigned long long)));
  const unsigned long long one[4] = {1, 1, 1, 1};
  HIP_TRY(hipMemcpy(epoch, one, sizeof(one), hipMemcpyHostToDevice));
  p2pdev::Push T{};
  T.P.base[0] = (char*)mapped, T.P.base[1] = (char*)mapped;  // ("both ranks'" blocks are the owner's: rank 1 stores to rank 0 only, see lo / hi)
  T.world = 1, T.rank = 1, T.kind = 0, T.dst_off = 0, T.flag_off = (size_t)count * 8, T.epoch = epoch;
  for (int q = 0; q < 16; ++q) T.lo[q] = 0, T.hi[q] = count;
  hipStream_t st;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  k_ipc_store<<<std::max(1, std::min((count + 255) / 256, 256)), 256, 0, st>>>(T, count, seed);
  // raise() writes flag[kind * world + rank] for q < world: world = 2 lays the flags out as the owner reads them, one destination
  p2pdev::Push R = T;
  R.world = 2;
  R.P.base[1] = R.P.base[0];
  k_ipc_raise<<<1, 64, 0, st>>>(R);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipStreamDestroy(st);
  (void)hipFree(epoch);
  HIP_TRY(hipIpcCloseMemHandle(mapped));
  return 0;
}
// the OWNER again: raises its own flag, waits for both with the transport's wait_flags, compares the payload; returns the number of
// wrong entries (>= 2^20: the wait ran out of patience), negative on a HIP error.  Frees the block.
int pdlpdev_debug_ipc_wait(int device, void* base, int count, double seed)
{
  HIP_TRY(hipSetDevice(device));
  unsigned long long* flags = reinterpret_cast<unsigned long long*>((char*)base + (size_t)count * 8);
  unsigned long long* epoch = nullptr;
  int* bad = nullptr;
  HIP_TRY(hipMalloc((void**)&epoch, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc((void**)&bad, sizeof(int)));
  const unsigned long long one[4] = {1, 1, 1, 1};
  HIP_TRY(hipMemcpy(epoch, one, sizeof(one), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(bad, 0, sizeof(int)));
  const unsigned long long mine = 1;
  HIP_TRY(hipMemcpy(flags + 0, &mine, sizeof(mine), hipMemcpyHostToDevice));  // rank 0's own flag of exchange 0
  k_ipc_wait_and_check<<<1, 256>>>((const double*)base, flags, epoch, count, seed, bad);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  int h = 0;
  HIP_TRY(hipMemcpy(&h, bad, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(epoch), (void)hipFree(bad), (void)hipFree(base);
  return h;
}
}  // extern "C"
