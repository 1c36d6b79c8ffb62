// Host side of the sharded E-step's cross-GPU exchange: every rank owns one small inbox in uncached
// device memory; the peers map it through an IPC handle and the E-step kernel's reducers write their
// per-node totals straight into it over xGMI (no collective library call on the path).  The handles
// travel by whatever the host program has (torch.distributed.all_gather_object in rlvi_amd.dist).
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "rlvi_common.h"

using namespace rlvi;

// Host-side record of the workspaces whose peer table has been written: rlvi_estep_sharded_f32 refuses
// (RLVI_E_WS) a workspace that never saw rlvi_workspace_set_peers instead of launching on a table of
// whatever the memory held.
namespace {
std::mutex g_mu;
std::unordered_map<const void *, int> g_world;
}  // namespace
namespace rlvi {
int peers_world_of(const void *ws) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_world.find(ws);
    return it == g_world.end() ? 0 : it->second;
}
// rlvi_workspace_init: whatever lived at this address before is gone, and so is its peer table
void peers_forget(const void *ws) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_world.erase(ws);
}
}  // namespace rlvi

static_assert(sizeof(hipIpcMemHandle_t) == RLVI_PEER_HANDLE_BYTES, "IPC handle size");

extern "C" size_t rlvi_peer_inbox_bytes(void) { return PEER_INBOX_BYTES; }

// Allocate (zeroed) an inbox on the current device.  Uncached: written by other GPUs while this one
// polls it, so no cache on either side may hold a line of it.
extern "C" int rlvi_peer_alloc(void **inbox) {
    if (!inbox) return RLVI_E_NULL;
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, PEER_INBOX_BYTES, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(p, 0, PEER_INBOX_BYTES);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    *inbox = p;
    return 0;
}

extern "C" int rlvi_peer_free(void *inbox) {
    if (!inbox) return 0;
    return (int)hipFree(inbox);
}

extern "C" int rlvi_peer_export(void *inbox, void *handle64) {
    if (!inbox || !handle64) return RLVI_E_NULL;
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, inbox);
    if (e != hipSuccess) return (int)e;
    memcpy(handle64, &h, sizeof(h));
    return 0;
}

extern "C" int rlvi_peer_open(const void *handle64, void **inbox) {
    if (!handle64 || !inbox) return RLVI_E_NULL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    *inbox = p;
    return 0;
}

// 1 if the current device can map memory of device `peer_device` (ordinals of THIS process), 0 if not,
// negative on an error: asked before opening a handle that lives on another GPU.
extern "C" int rlvi_peer_can_access(int peer_device) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return RLVI_E_WS;
    if (peer_device == cur) return 1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || peer_device < 0 || peer_device >= n) return RLVI_E_SHAPE;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, cur, peer_device) != hipSuccess) return RLVI_E_WS;
    return can ? 1 : 0;
}

extern "C" int rlvi_peer_close(void *inbox) {
    if (!inbox) return 0;
    return (int)hipIpcCloseMemHandle(inbox);
}

// Write the peer table of a workspace: inboxes[r] = rank r's inbox as mapped in this process
// (inboxes[rank] = the local allocation).  Resets the round counter: every rank must call this at
// the same point of the program.  world == 1 (or inboxes == NULL) clears the table.
extern "C" int rlvi_workspace_set_peers(void *ws, int rank, int world, void *const *inboxes, void *stream) {
    if (!ws) return RLVI_E_NULL;
    if (world < 1 || world > MAX_PEERS || rank < 0 || rank >= world) return RLVI_E_SHAPE;
    if (world > 1 && !inboxes) return RLVI_E_NULL;
    PeerTable t;
    memset(&t, 0, sizeof(t));
    t.world = world;
    t.rank = rank;
    for (int r = 0; r < world && inboxes; ++r) {
        if (world > 1 && !inboxes[r]) return RLVI_E_NULL;
        t.inbox[r] = (unsigned long long)(uintptr_t)inboxes[r];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the sharded solves start cold, and alike on every rank
    hipError_t e = hipMemsetAsync(static_cast<char *>(ws) + WS_PEER_OFF, 0, WS_PEER_BYTES, st);
    if (e != hipSuccess) return (int)e;
    // ... and with a clean inbox: the round counter restarts at 0, so records an earlier set-up left in
    // the inbox (tags 1..n) would pass for the new rounds' records.  (The caller's barrier over the ranks
    // between this call and the first sharded call keeps a peer's push from racing this memset.)
    if (inboxes && inboxes[rank]) {
        e = hipMemsetAsync(inboxes[rank], 0, PEER_INBOX_BYTES, st);
        if (e != hipSuccess) return (int)e;
    }
    e = hipMemcpyAsync(static_cast<char *>(ws) + WS_PEER_OFF, &t, sizeof(t), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lk(g_mu);
    g_world[ws] = world;
    return 0;
}

// Forget the peer table of a workspace (before the inboxes it points at are unmapped / freed): sharded calls
// on it return RLVI_E_WS from then on instead of launching on stale addresses.
extern "C" int rlvi_workspace_clear_peers(void *ws, void *stream) {
    if (!ws) return RLVI_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(static_cast<char *>(ws) + WS_PEER_OFF, 0, WS_PEER_BYTES, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);
    peers_forget(ws);
    return (int)e;
}
