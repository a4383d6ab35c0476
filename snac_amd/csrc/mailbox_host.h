// mailbox_host.h -- the mailbox of the resident stepper (k_mailbox.hip) and the HOST half of its protocol.
//
// Included by exactly one translation unit per binary: k_mailbox.hip (the product: the hooks below launch the real kernel) and
// tests/native/mailbox_host_test.cpp (gcc, -fsanitize=address,undefined / thread, against tests/native/fakehip: the "wave" is a host
// thread that speaks the device half of the protocol).  Nothing in here is device code; the HIP calls it makes are
// hipHostMalloc / hipHostFree, hipStreamCreateWithFlags / Destroy / Synchronize / Query, hipGetDevice / hipSetDevice.
//
// Protocol (one mailbox = a batch of 1 .. 256 envs = 1 .. 4 resident wavefronts, wave w owns envs [64 w, 64 w + 64); the waves are
// the blocks of ONE launch on the mailbox's own stream -- a launch per wave on a stream per wave was tried first: the fourth stream shares
// a hardware queue with another and its wave starts only when that one's has left, 1 ms per step, profiles/r06_mailbox.txt):
//   host -> device   `cmd`: ONE 8-byte word stored atomically: bits 0-31 sequence number | 32-39 op | 40-47 action | 48-51 step size |
//                    52-63 state generation; a batch's actions / step sizes are written BEFORE it
//   device -> host   wave w: rows, rewards, done flags, then ack_seq[w] = the sequence number (release); the write-through of the
//                    records to HBM trails the acknowledgement and is reported in wt_seq[w]; alive[w] = 0 is a wave's last store
//   a wave leaves    on MB_QUIT, or after idle_us without a command -- an exit every wave reaches
//   cancelling       a command whose acknowledgement does not arrive (snac_mailbox_step: the launch is still QUEUED behind other work
//                    after SNAC_MAILBOX_TIMEOUT_S seconds) is WITHDRAWN before the error is returned: the command word is replaced by
//                    MB_QUIT with the same sequence number, so a wave that starts later leaves without stepping (it reports that in
//                    quit_seq[w]); if the step was served after all in that window the call returns SNAC_OK.  ADVICE round 5: a stale
//                    step executed behind an exception the caller already got.
#pragma once
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "snac_common.h"

constexpr int MB_MAX_WAVES = 4, MB_WAVE_ENVS = 64, MB_MAX_ENVS = MB_MAX_WAVES * MB_WAVE_ENVS;
enum { MB_STEP = 1, MB_STEP_N = 2, MB_QUIT = 3 };   // MB_STEP_N: snac_mailbox_step_n -- reward / done always go to their own arrays too

struct snac_mailbox {
    // ---- host -> device (its own cache line, written by the host only).  The whole command is ONE 8-byte word: every word the wave
    // would have to fetch separately is one more round trip over the bus (a first version with four words: 10.4 us per step)
    uint64_t cmd;
    uint32_t pad0[14];
    // a batch of 2 .. 256 envs (one env per lane): its actions and step sizes, written by the host BEFORE the command word (which then
    // carries no action); a wave fetches its 64 + 64 bytes with one trip
    int8_t actions[MB_MAX_ENVS];
    int8_t steps[MB_MAX_ENVS];
    // ---- device -> host: ONE line the host polls, a word per wave and kind
    uint32_t ack_seq[MB_MAX_WAVES];        // = the command's sequence number once wave w's rows are complete
    uint32_t alive[MB_MAX_WAVES];          // 1 while wave w is resident (set by the host before the launch), 0 stored as the wave's last act
    uint32_t wt_seq[MB_MAX_WAVES];         // = ack_seq once the wave's records in HBM are complete too (written behind the acknowledgement)
    uint32_t quit_seq[MB_MAX_WAVES];       // the sequence number of the last MB_QUIT wave w obeyed (tells a withdrawn step from a served one)
    // ---- device -> host, statistics (another line)
    uint32_t steps_served[MB_MAX_WAVES];
    uint32_t dbg[4];                       // wave 0: ticks of the 100 MHz clock of the last step: stepped, rows stored, fenced, write-through
    uint32_t pad1[8];
    float reward[MB_MAX_ENVS];             // per env, written with the rows
    uint8_t done[MB_MAX_ENVS];
    // ---- host side bookkeeping (never read by the device)
    hipStream_t stream;                    // its own queue: resident waves must not sit in front of anyone's work
    uint32_t req_seq, state_gen;           // the host's own copies of what it last posted
    int32_t armed;                         // a launch (all waves) has been made and not yet seen to end
    int32_t launches;
    int32_t row_values;                    // values per row
    int32_t device;                        // the device the mailbox was created on: every launch and stream call happens under it
    uint32_t idle_us;
    int32_t num_envs, num_waves;
    uint32_t timeout_ms;                   // snac_mailbox_step: how long a QUEUED launch is waited for before the command is withdrawn
    uint32_t pad2[4];
    double row[512];                       // the observation rows [num_envs][row_values] of obs_dtype, written by the device (the allocation extends past 512 values when they need it)
};
static_assert(offsetof(snac_mailbox, actions) == 64 && offsetof(snac_mailbox, ack_seq) == 576 && offsetof(snac_mailbox, steps_served) == 640 &&
              offsetof(snac_mailbox, reward) % 64 == 0 && offsetof(snac_mailbox, row) % 64 == 0, "mailbox cache lines");

namespace snac_mb {

// ---- what the including translation unit provides ------------------------------------------------------------------------------
int hook_check_desc(const snac_env_desc* d, int* row_values);                   // the descriptor is valid for a mailbox; values per row
int hook_check_state(const snac_env_desc* d, const snac_state* st);            // the state's pointers are usable (check_common)
int hook_launch(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st);   // k_mailbox, a block per wave, on mb->stream

using snac_detail::fail;
using snac_detail::fail_hip;

inline uint32_t host_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void host_store(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}

// the mailbox's device current for the scope (a facade built on cuda:1 while cuda:0 is current must not put its wave on GPU 0: ADVICE round 5)
struct DeviceScope {
    int prev = -1, dev;
    explicit DeviceScope(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

inline uint64_t command_word(uint32_t seq, int op, int action, int step_size, uint32_t gen) {
    return (uint64_t)seq | ((uint64_t)(op & 0xff) << 32) | ((uint64_t)(uint8_t)(int8_t)action << 40) | ((uint64_t)(step_size & 0xf) << 48) |
           ((uint64_t)(gen & 0xfffu) << 52);
}
inline void post(snac_mailbox* mb, int op, int action, int step_size) {
    mb->req_seq += 1u;
    __atomic_store_n(&mb->cmd, command_word(mb->req_seq, op, action, step_size, mb->state_gen), __ATOMIC_RELEASE);
}
inline bool any_alive(const snac_mailbox* mb) {
    for (int w = 0; w < mb->num_waves; ++w) if (host_load(&mb->alive[w])) return true;
    return false;
}
inline bool all_acked(const snac_mailbox* mb, uint32_t req) {
    for (int w = 0; w < mb->num_waves; ++w) if (host_load(&mb->ack_seq[w]) != req) return false;
    return true;
}

// (Re)launch the waves.  The previous launch, if any, is retired first: waves that are still resident leave after idle_us without a
// command (one that is pending they serve first), so this waits a millisecond at most -- the price of the rare case that a command
// arrives while SOME waves of a batch have just left on their idle clocks.
inline int arm(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st) {
    DeviceScope scope(mb->device);
    if (mb->armed) {
        const hipError_t e = hipStreamSynchronize(mb->stream);
        if (e != hipSuccess) return fail_hip(e, "mailbox stream");
        mb->armed = 0;
    }
    for (int w = 0; w < mb->num_waves; ++w) host_store(&mb->alive[w], 1u);   // (a wave clears its flag as its last act)
    if (int rc = hook_launch(mb, d, st)) {
        for (int w = 0; w < mb->num_waves; ++w) host_store(&mb->alive[w], 0u);
        return rc;
    }
    mb->armed = 1;
    mb->launches += 1;
    return SNAC_OK;
}

// The command cannot be served (in time): take it back.  Returns the set of waves (bit w) that turn out to have served it after all
// (all of them: the step counts as served; some of them: a batch of several waves whose launch was cut in two by the limit -- the caller
// is told which envs stepped).  Leaves the error string alone.
inline uint32_t withdraw(snac_mailbox* mb, uint32_t req) {
    __atomic_store_n(&mb->cmd, command_word(req, MB_QUIT, 0, 1, mb->state_gen), __ATOMIC_RELEASE);
    // a wave that had fetched the step before the word changed is serving it right now (microseconds); one that starts later obeys the QUIT
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50)) {
        if (all_acked(mb, req)) break;
        cpu_relax();
    }
    uint32_t served = 0;
    for (int w = 0; w < mb->num_waves; ++w)
        if (host_load(&mb->ack_seq[w]) == req && host_load(&mb->quit_seq[w]) != req) served |= 1u << w;
    return served;
}
inline uint32_t all_waves(const snac_mailbox* mb) { return (1u << mb->num_waves) - 1u; }

// a wave that is not resident and has not acknowledged `req`: it left (idle timeout) before it saw the command -- or none was ever launched
inline bool wave_missing(const snac_mailbox* mb, uint32_t req) {
    for (int w = 0; w < mb->num_waves; ++w) {
        if (host_load(&mb->ack_seq[w]) == req || host_load(&mb->alive[w])) continue;
        if (host_load(&mb->ack_seq[w]) != req) return true;          // (alive == 0 was the wave's LAST store: its acknowledgement is final now)
    }
    return false;
}

inline int await_ack(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st) {
    const uint32_t req = mb->req_seq;
    bool checked = false;
    auto rearm = [&]() -> int {
        if (!checked) {
            if (int rc = hook_check_state(d, st)) return rc;
            if (d->num_envs != mb->num_envs) return fail(SNAC_ERR_ARG, "mailbox of another batch size");
            checked = true;
        }
        return arm(mb, d, st);                                       // new waves find req != their acknowledgement and serve it
    };
    if (wave_missing(mb, req))
        if (int rc = rearm()) { (void)withdraw(mb, req); return rc; }
    const auto t0 = std::chrono::steady_clock::now();
    const auto check_every = std::chrono::milliseconds(mb->timeout_ms < 2000u ? mb->timeout_ms : 2000u);
    auto next_check = check_every;
    for (unsigned spins = 0;; ++spins) {
        if (all_acked(mb, req)) return SNAC_OK;
        if (wave_missing(mb, req))
            if (int rc = rearm()) { (void)withdraw(mb, req); return rc; }
        cpu_relax();
        if ((spins & 0xFFFFu) != 0xFFFFu) continue;
        const auto el = std::chrono::steady_clock::now() - t0;
        if (el < next_check) continue;
        next_check += check_every;
        // No acknowledgement for seconds.  A wave serves a command within microseconds of seeing it, so the launch has not STARTED: it is
        // queued behind kernels that fill the device (long rollouts, a trainer's kernels) -- not a failure.  Keep waiting while the
        // launch is pending; fail on a real error, or when the limit is reached -- and then WITHDRAW the command first.
        {
            DeviceScope scope(mb->device);
            const hipError_t q = hipStreamQuery(mb->stream);
            if (q != hipErrorNotReady && q != hipSuccess) { (void)withdraw(mb, req); return fail_hip(q, "mailbox stream"); }
            if (q == hipSuccess) {
                // the stream is idle: every wave has ended.  Each clears its flag as its last act, so wave_missing() relaunches -- unless
                // one died without doing so
                for (int w = 0; w < mb->num_waves; ++w)
                    if (host_load(&mb->alive[w]) && host_load(&mb->ack_seq[w]) != req) host_store(&mb->alive[w], 0u);
            }
        }
        if (el > std::chrono::milliseconds(mb->timeout_ms)) {
            const uint32_t served = withdraw(mb, req);
            if (served == all_waves(mb)) return SNAC_OK;
            if (served == 0)
                return fail(SNAC_ERR_HIP, "mailbox: no acknowledgement in time (the waves' launch is still queued behind other work); the command was "
                                          "withdrawn, the state is as the last acknowledged step left it");
            char msg[200];
            std::snprintf(msg, sizeof msg, "mailbox: no acknowledgement from every wave in time; the command was withdrawn, but the envs of the waves in "
                                           "mask 0x%x (64 envs each, wave w = envs 64 w ..) HAVE taken the step", served);
            return fail(SNAC_ERR_HIP, msg);
        }
    }
}

}  // namespace snac_mb

// ---- the entry points (include/snac_hip.h; defined HERE: one definition per binary, this header has exactly one includer) --------
extern "C" {

double* snac_mailbox_row(snac_mailbox* mb) { return mb ? mb->row : nullptr; }
float* snac_mailbox_reward(snac_mailbox* mb) { return mb ? mb->reward : nullptr; }
uint8_t* snac_mailbox_done(snac_mailbox* mb) { return mb ? mb->done : nullptr; }

int snac_mailbox_create(const snac_env_desc* d, uint32_t idle_us, snac_mailbox** out) {
    using namespace snac_mb;
    if (!d || !out) return fail(SNAC_ERR_ARG, "null desc / out");
    *out = nullptr;
    if (d->num_envs < 1 || d->num_envs > MB_MAX_ENVS)
        return fail(SNAC_ERR_UNSUPPORTED, "the mailbox steps a batch of 1 .. 256 envs (up to four resident wavefronts, an env per lane)");
    int ld = 0;
    if (int rc = hook_check_desc(d, &ld)) return rc;
    const size_t row_bytes = (size_t)d->num_envs * (size_t)ld * sizeof(double);
    const size_t own = sizeof(((snac_mailbox*)nullptr)->row);
    const size_t total = sizeof(snac_mailbox) + (row_bytes > own ? row_bytes - own : 0);
    snac_mailbox* mb = nullptr;
    // coherent (fine-grained) page-locked memory, mapped: device stores and host stores are visible to the other side while the
    // wave runs -- what the doorbell and the acknowledgement need (plain pinned memory is only guaranteed at kernel boundaries)
    hipError_t e = hipHostMalloc((void**)&mb, total, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return fail_hip(e, "hipHostMalloc(mailbox)");
    std::memset(mb, 0, total);
    mb->num_envs = d->num_envs;
    mb->num_waves = (d->num_envs + MB_WAVE_ENVS - 1) / MB_WAVE_ENVS;
    e = hipStreamCreateWithFlags(&mb->stream, hipStreamNonBlocking);  // its own queue: resident waves must not sit in front of anyone's work
    if (e != hipSuccess) { (void)hipHostFree(mb); return fail_hip(e, "hipStreamCreate(mailbox)"); }
    mb->row_values = ld;
    mb->idle_us = idle_us ? idle_us : 1000;
    const char* ts = std::getenv("SNAC_MAILBOX_TIMEOUT_S");
    const double tsec = ts ? std::atof(ts) : 120.0;
    mb->timeout_ms = (uint32_t)((tsec > 0.001 ? (tsec < 86400.0 ? tsec : 86400.0) : 0.001) * 1000.0);
    if (hipGetDevice(&mb->device) != hipSuccess) mb->device = 0;     // the caller's current device: the one its state lives on
    *out = mb;
    return SNAC_OK;
}

int snac_mailbox_touch(snac_mailbox* mb) {
    if (!mb) return snac_detail::fail(SNAC_ERR_ARG, "null mailbox");
    mb->state_gen += 1u;                                             // travels with the next command
    return SNAC_OK;
}

int snac_mailbox_step(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st, int32_t action, int32_t step_size) {
    using namespace snac_mb;
    if (!mb || !d || !st) return fail(SNAC_ERR_ARG, "null mailbox / desc / state");
    if (mb->num_envs != 1 || d->num_envs != 1) return fail(SNAC_ERR_ARG, "snac_mailbox_step is for a batch of one env (snac_mailbox_step_n)");
    const int lim = d->kind == SNAC_ENV_1D ? 3 : (d->kind == SNAC_ENV_2D ? 5 : 8);
    post(mb, MB_STEP, (action >= 0 && action < lim) ? action : -1, step_size < 1 ? 1 : (step_size > 3 ? 3 : step_size));   // (every invalid action steps alike)
    return await_ack(mb, d, st);
}

int snac_mailbox_step_n(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st, const int8_t* actions, const int8_t* step_size) {
    using namespace snac_mb;
    if (!mb || !d || !st || !actions || !step_size) return fail(SNAC_ERR_ARG, "null mailbox / desc / state / actions / step_size");
    if (d->num_envs != mb->num_envs) return fail(SNAC_ERR_ARG, "mailbox of another batch size");   // (before anything is read by its size)
    std::memcpy(mb->actions, actions, (size_t)mb->num_envs);
    std::memcpy(mb->steps, step_size, (size_t)mb->num_envs);
    post(mb, MB_STEP_N, actions[0], step_size[0] < 1 ? 1 : (step_size[0] > 3 ? 3 : step_size[0]));
    return await_ack(mb, d, st);
}

// Waits until the records in HBM hold the last acknowledged step (the waves write them through BEHIND their acknowledgement): what an
// entry point that reads or changes them calls first.  A microsecond at most; SNAC_ERR_HIP after 2 s.
int snac_mailbox_settle(snac_mailbox* mb) {
    using namespace snac_mb;
    if (!mb) return SNAC_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (!mb->armed) return SNAC_OK;
    for (int w = 0; w < mb->num_waves; ++w) {
        const uint32_t want = host_load(&mb->ack_seq[w]);
        for (unsigned spins = 0; host_load(&mb->wt_seq[w]) != want; ++spins) {
            if (!host_load(&mb->alive[w]) && host_load(&mb->wt_seq[w]) == host_load(&mb->ack_seq[w])) break;
            cpu_relax();
            if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
                return fail(SNAC_ERR_HIP, "mailbox: write-through not seen within 2 s");
        }
    }
    return SNAC_OK;
}

// Ends the resident waves (if any) and waits until they are gone.
int snac_mailbox_quit(snac_mailbox* mb) {
    using namespace snac_mb;
    if (!mb) return SNAC_OK;
    if (!mb->armed) return SNAC_OK;
    if (any_alive(mb)) post(mb, MB_QUIT, 0, 1);
    DeviceScope scope(mb->device);
    int rc = SNAC_OK;
    const hipError_t e = hipStreamSynchronize(mb->stream);           // (idle waves end by themselves within idle_us)
    mb->armed = 0;
    if (e != hipSuccess) rc = fail_hip(e, "mailbox stream");
    for (int w = 0; w < mb->num_waves; ++w) host_store(&mb->alive[w], 0u);
    for (int w = 0; w < mb->num_waves; ++w) {                        // nothing is pending for the next waves
        host_store(&mb->ack_seq[w], mb->req_seq);
        host_store(&mb->wt_seq[w], mb->req_seq);
    }
    return rc;
}

int snac_mailbox_destroy(snac_mailbox* mb) {
    if (!mb) return SNAC_OK;
    const int rc = snac_mailbox_quit(mb);
    snac_mb::DeviceScope scope(mb->device);
    (void)hipStreamDestroy(mb->stream);
    (void)hipHostFree(mb);
    return rc;
}

// {launches, steps served by resident waves (wave 0's count: every wave serves every command), a wave is resident, idle_us, and of wave
// 0's last step, in ticks of the GPU's 100 MHz clock: step, row stores issued, fence before the acknowledgement, write-through behind it}
int snac_mailbox_stats(const snac_mailbox* mb, uint32_t out[8]) {
    using namespace snac_mb;
    if (!mb || !out) return fail(SNAC_ERR_ARG, "null mailbox / out");
    out[0] = (uint32_t)mb->launches; out[1] = host_load(&mb->steps_served[0]); out[2] = any_alive(mb) ? 1u : 0u; out[3] = mb->idle_us;
    for (int i = 0; i < 4; ++i) out[4 + i] = host_load(&mb->dbg[i]);
    return SNAC_OK;
}

}  // extern "C"

