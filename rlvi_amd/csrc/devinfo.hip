// Host-side device facts and tuning knobs shared by the launchers.
//
//   * device_info(): CU count of the current device, asked from the runtime once per device --
//     the launch geometry of every kernel is derived from it, never from a constant.
//   * coop_blocks(): how many workgroups of a kernel are provably co-resident (occupancy query
//     x CU count, with the margin MI355X_MICROARCH.md prescribes where the API over-reports);
//     every kernel whose workgroups wait for each other sizes its exchanging grid with it.
//   * tune_get()/rlvi_tune_set(): integer knobs, default <- environment <- rlvi_tune_set().
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "rlvi_common.h"

namespace rlvi {

namespace {
struct Knob {
    char name[48];
    int value;
    bool from_set;
};
constexpr int MAX_KNOBS = 96;
Knob g_knobs[MAX_KNOBS];
int g_nknobs = 0;
std::mutex g_mu;

Knob *find(const char *name) {
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].name, name) == 0) return &g_knobs[i];
    return nullptr;
}
Knob *add(const char *name, int value, bool from_set) {
    if (g_nknobs >= MAX_KNOBS || strlen(name) >= sizeof(g_knobs[0].name)) return nullptr;
    Knob *k = &g_knobs[g_nknobs++];
    strcpy(k->name, name);
    k->value = value;
    k->from_set = from_set;
    return k;
}

constexpr int MAX_DEV = 64;
DeviceInfo g_dev[MAX_DEV];
bool g_dev_ok[MAX_DEV];
}  // namespace

int tune_get(const char *name, int dflt) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (Knob *k = find(name)) return k->value;
    const char *e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    if (e) add(name, v, false);      // defaults are not cached: a later rlvi_tune_set still wins
    return v;
}

const DeviceInfo &device_info() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_dev_ok[dev]) {
        DeviceInfo d;
        d.cus = NUM_CU_DEFAULT;
        d.lds_per_cu = 160 * 1024;
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            d.cus = v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess &&
            v > 0)
            d.lds_per_cu = v;
        g_dev[dev] = d;
        g_dev_ok[dev] = true;
    }
    return g_dev[dev];
}

int coop_blocks_from_occupancy(int per_cu_api, int block_threads, int cus) {
    // MI355X_MICROARCH.md, "Residency and cooperative launch": the occupancy API can answer one
    // block per CU too many for 256-thread blocks with 81..112 SGPRs, and neither the plain nor the
    // graph launch path checks.  One block per CU of margin where more than one is promised.
    int per_cu = per_cu_api;
    if (per_cu > 1 && block_threads <= 256) per_cu -= 1;
    if (per_cu < 0) per_cu = 0;
    long long n = (long long)per_cu * cus;
    return n > (1 << 20) ? (1 << 20) : (int)n;
}

namespace {
struct CapEntry { const void *kernel; int dev, block; size_t lds; int cap; };
constexpr int MAX_CAPS = 256;
CapEntry g_caps[MAX_CAPS];
int g_ncaps = 0;
}  // namespace

int coop_cap_cached(const void *kernel, int block_threads, size_t dyn_lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    // debugging / tests (RLVI_COOP_CAP): pretend the device admits fewer co-resident workgroups
    const int forced = tune_get("RLVI_COOP_CAP", 0);
    int cap = -1;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (int i = 0; i < g_ncaps; ++i)
            if (g_caps[i].kernel == kernel && g_caps[i].dev == dev && g_caps[i].block == block_threads &&
                g_caps[i].lds == dyn_lds)
                cap = g_caps[i].cap;
    }
    if (cap < 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dyn_lds) != hipSuccess)
            per_cu = 0;
        cap = coop_blocks_from_occupancy(per_cu, block_threads, device_info().cus);
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_ncaps < MAX_CAPS) g_caps[g_ncaps++] = CapEntry{kernel, dev, block_threads, dyn_lds, cap};
    }
    // Several PROCESSES on this device (rlvi_amd.dist.declare_device_sharing finds them by PCI bus id, or
    // the environment says so): the occupancy query above sees one process's kernels only, so S processes
    // that each size a cooperating grid from it can together ask for more workgroups than the device holds
    // -- each grid then waits for slots the other one occupies until the spin bound ends both (the
    // RLVI_ST_TIMEOUT recorded in round 2: two ranks on one GPU, each with 256 workgroups of a kernel the
    // device admits two of per CU).  Every process takes 1/S of the proven capacity instead.
    const int sharers = tune_get("RLVI_DEVICE_SHARERS", 1);
    if (sharers > 1) cap /= sharers;
    return (forced > 0 && forced < cap) ? forced : cap;
}

}  // namespace rlvi

extern "C" int rlvi_tune_set(const char *name, int value) {
    if (!name) return RLVI_E_NULL;
    std::lock_guard<std::mutex> lk(rlvi::g_mu);
    if (rlvi::Knob *k = rlvi::find(name)) {
        k->value = value;
        k->from_set = true;
        return 0;
    }
    return rlvi::add(name, value, true) ? 0 : RLVI_E_LIMIT;
}

// Forget a value set by rlvi_tune_set: the knob is its environment variable / built-in default again.
// Returns 1 if there was such a value, 0 if not.
extern "C" int rlvi_tune_unset(const char *name) {
    if (!name) return RLVI_E_NULL;
    std::lock_guard<std::mutex> lk(rlvi::g_mu);
    rlvi::Knob *k = rlvi::find(name);
    if (!k || !k->from_set) return 0;
    *k = rlvi::g_knobs[--rlvi::g_nknobs];       // (order is irrelevant)
    return 1;
}

// Names of the knobs that currently carry a rlvi_tune_set value, comma-separated, into buf[len] (cut at len - 1);
// returns how many there are.  A test harness asserts 0 after every test: a knob is process-wide, and one that
// a test forgot to take back changes which kernel every later call runs.
extern "C" int rlvi_tune_overrides(char *buf, int len) {
    std::lock_guard<std::mutex> lk(rlvi::g_mu);
    int n = 0;
    size_t pos = 0;
    if (buf && len > 0) buf[0] = 0;
    for (int i = 0; i < rlvi::g_nknobs; ++i) {
        if (!rlvi::g_knobs[i].from_set) continue;
        ++n;
        if (!buf || len <= 0) continue;
        const size_t l = strlen(rlvi::g_knobs[i].name);
        if (pos + l + 2 < (size_t)len) {
            if (pos) buf[pos++] = ',';
            memcpy(buf + pos, rlvi::g_knobs[i].name, l);
            pos += l;
            buf[pos] = 0;
        }
    }
    return n;
}

// ---- per-workspace launch options (host side: what a launcher needs to know about the CALLER of this workspace,
// as opposed to the process-wide tuning knobs above)
namespace rlvi {
namespace {
struct WsOptions { int v[WSOPT_COUNT]; bool set[WSOPT_COUNT]; int last_mstep; };
std::unordered_map<const void *, WsOptions> g_wsopt;
std::mutex g_wsopt_mu;
}  // namespace
int ws_option(const void *ws, int which, int dflt) {
    if (which < 0 || which >= WSOPT_COUNT) return dflt;
    std::lock_guard<std::mutex> lk(g_wsopt_mu);
    auto it = g_wsopt.find(ws);
    return (it != g_wsopt.end() && it->second.set[which]) ? it->second.v[which] : dflt;
}
void ws_note_mstep(const void *ws, int code) {
    std::lock_guard<std::mutex> lk(g_wsopt_mu);
    g_wsopt[ws].last_mstep = code;
}
void ws_options_forget(const void *ws) {
    std::lock_guard<std::mutex> lk(g_wsopt_mu);
    g_wsopt.erase(ws);
}
}  // namespace rlvi

extern "C" int rlvi_workspace_set_option(void *ws, const char *name, int value) {
    if (!ws || !name) return RLVI_E_NULL;
    int which = -1;
    if (strcmp(name, "logits_from_hbm") == 0) which = rlvi::WSOPT_LOGITS_FROM_HBM;
    else if (strcmp(name, "cold_start") == 0) which = rlvi::WSOPT_COLD_START;
    if (which < 0) return RLVI_E_SHAPE;
    std::lock_guard<std::mutex> lk(rlvi::g_wsopt_mu);
    rlvi::WsOptions &o = rlvi::g_wsopt[ws];
    o.v[which] = value;
    o.set[which] = true;
    return 0;
}

// Which form the LAST M-step launch on this workspace took (for tests of the dispatch): 0 = none yet, 1 = register
// rows, 2 = wave tiles in four-wave workgroups, 3 = wave tiles in 16-wave workgroups (barrier behind the load
// issue); + 16 when the launch carried a timed hold (the caller's "logits_from_hbm" hint or the lab knob).
extern "C" int rlvi_workspace_last_mstep_form(const void *ws) {
    std::lock_guard<std::mutex> lk(rlvi::g_wsopt_mu);
    auto it = rlvi::g_wsopt.find(ws);
    return it == rlvi::g_wsopt.end() ? 0 : it->second.last_mstep;
}

extern "C" int rlvi_device_cus(void) { return rlvi::device_info().cus; }

// "0000:c1:00.0"-style PCI bus id of the current device: the same physical GPU has the same id in every
// process of the node whatever HIP_VISIBLE_DEVICES made of the ordinals, so ranks that compare ids know
// whether they share a device (rlvi_amd.dist.declare_device_sharing -> RLVI_DEVICE_SHARERS).
extern "C" int rlvi_device_pci_bus_id(char *buf, int len) {
    if (!buf) return RLVI_E_NULL;
    if (len < 16) return RLVI_E_SHAPE;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceGetPCIBusId(buf, len, dev);
}
