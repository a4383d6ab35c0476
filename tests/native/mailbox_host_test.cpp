// mailbox_host_test.cpp -- the HOST half of the resident-stepper protocol (snac_amd/csrc/mailbox_host.h, the very header k_mailbox.hip
// compiles) against tests/native/fakehip: the "wavefront" is a host thread that speaks the device half of the protocol, the stream a
// worker thread.  Built by tests/test_native_host_logic.py with gcc -fsanitize=address,undefined and again with -fsanitize=thread.
// Test infrastructure.  Cases: argument errors, allocation failures at every call, 20 000 steps, idle exit and re-arm, four waves / 256
// envs, a launch that is QUEUED for seconds (no failure), the hard limit (the command is WITHDRAWN: a wave that starts later must not
// step), a step served while it is being withdrawn, state generations, quit / destroy with a resident wave, another current device.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mailbox_host.h"

namespace snac_detail { thread_local char g_err[256] = ""; }
extern "C" const char* snac_last_error(void) { return snac_detail::g_err; }

// ---- the fake device side ----------------------------------------------------------------------------------------------------------
static constexpr int LD = 4;                                         // values per row: {steps taken by this env, action, step size, env index}
static std::atomic<int> g_launch_device{-1}, g_reloads{0}, g_hold_ms{0}, g_step_ms{0}, g_launch_fail{0};
static std::atomic<long long> g_env_steps[MB_MAX_ENVS];
static std::atomic<int> g_wave_late_ms[MB_MAX_WAVES];                 // wave w of the NEXT launches starts this late (a launch cut in two)

template <typename T> static T dload(const T* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
template <typename T> static void dstore(T* p, T v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

static void fake_wave(snac_mailbox* mb, int wv) {
    using clk = std::chrono::steady_clock;
    const int env0 = wv * MB_WAVE_ENVS, nenv = std::min(MB_WAVE_ENVS, mb->num_envs - env0);
    uint32_t seen = dload(&mb->ack_seq[wv]), served = dload(&mb->steps_served[wv]), gen = 0xFFFFFFFFu;
    bool quit = false, wt_pending = false;
    if (const int late = g_wave_late_ms[wv].load()) std::this_thread::sleep_for(std::chrono::milliseconds(late));
    auto last = clk::now();
    for (;;) {
        const uint64_t cmd = __atomic_load_n(&mb->cmd, __ATOMIC_ACQUIRE);
        if (wt_pending) { dstore(&mb->steps_served[wv], served); dstore(&mb->wt_seq[wv], seen); wt_pending = false; }
        const uint32_t req = (uint32_t)cmd;
        if (req == seen) {
            if (clk::now() - last > std::chrono::microseconds(mb->idle_us)) break;
            std::this_thread::yield();
            continue;
        }
        const int op = (int)((cmd >> 32) & 0xffu);
        if (op == MB_QUIT) { seen = req; quit = true; break; }
        const uint32_t g = (uint32_t)(cmd >> 52);
        if (g != gen) { gen = g; ++g_reloads; }
        if (g_step_ms.load()) std::this_thread::sleep_for(std::chrono::milliseconds(g_step_ms.load()));   // a (very) slow step
        for (int e = 0; e < nenv; ++e) {
            const int env = env0 + e;
            int act = (int)(int8_t)((cmd >> 40) & 0xffu), k = (int)((cmd >> 48) & 0xfu);
            if (mb->num_envs > 1) { act = mb->actions[env]; k = mb->steps[env]; }
            const long long n = ++g_env_steps[env];
            double* row = mb->row + (size_t)env * LD;
            row[0] = (double)n; row[1] = (double)act; row[2] = (double)k; row[3] = (double)env;
            mb->reward[env] = (float)act; mb->done[env] = (uint8_t)(n & 1);
        }
        dstore(&mb->ack_seq[wv], req);
        served += 1u; seen = req; wt_pending = true; last = clk::now();
    }
    dstore(&mb->steps_served[wv], served);
    if (quit) dstore(&mb->quit_seq[wv], seen);
    dstore(&mb->ack_seq[wv], seen);
    dstore(&mb->wt_seq[wv], seen);
    dstore(&mb->alive[wv], 0u);
}

namespace snac_mb {
int hook_check_desc(const snac_env_desc* d, int* row_values) {
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    *row_values = LD;
    return SNAC_OK;
}
int hook_check_state(const snac_env_desc*, const snac_state* st) { return st ? SNAC_OK : fail(SNAC_ERR_ARG, "null state"); }
int hook_launch(snac_mailbox* mb, const snac_env_desc*, const snac_state*) {
    if (g_launch_fail.load() > 0 && g_launch_fail.fetch_sub(1) == 1) return fail(SNAC_ERR_HIP, "injected launch failure");
    int dev = -1;
    (void)hipGetDevice(&dev);
    g_launch_device = dev;
    if (const int ms = g_hold_ms.exchange(0))                        // the launch sits behind someone's long kernel
        fakehip::enqueue(mb->stream, [ms] { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); });
    fakehip::enqueue(mb->stream, [mb] {                              // one "kernel": its blocks run side by side, the launch ends when all have
        std::vector<std::thread> blocks;
        for (int w = 0; w < mb->num_waves; ++w) blocks.emplace_back(fake_wave, mb, w);
        for (auto& t : blocks) t.join();
    });
    return SNAC_OK;
}
}  // namespace snac_mb

// ---- the cases ------------------------------------------------------------------------------------------------------------------------
static int g_failed = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s   (last error: %s)\n", __LINE__, #c, snac_last_error()); ++g_failed; } } while (0)

static snac_env_desc desc(int n) {
    snac_env_desc d;
    std::memset(&d, 0, sizeof(d));
    d.kind = SNAC_ENV_2D; d.num_envs = n;
    return d;
}
static void reset_counts() { for (auto& c : g_env_steps) c = 0; g_reloads = 0; }
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

static void case_arguments_and_allocation_failures() {
    snac_mailbox* mb = nullptr;
    snac_env_desc d = desc(1);
    CHECK(snac_mailbox_create(nullptr, 0, &mb) == SNAC_ERR_ARG);
    CHECK(snac_mailbox_create(&d, 0, nullptr) == SNAC_ERR_ARG);
    d.num_envs = 0;   CHECK(snac_mailbox_create(&d, 0, &mb) == SNAC_ERR_UNSUPPORTED && !mb);
    d.num_envs = 257; CHECK(snac_mailbox_create(&d, 0, &mb) == SNAC_ERR_UNSUPPORTED && !mb);
    d = desc(1); d.kind = 9; CHECK(snac_mailbox_create(&d, 0, &mb) == SNAC_ERR_ARG);
    CHECK(snac_mailbox_step(nullptr, &d, nullptr, 0, 1) == SNAC_ERR_ARG);
    CHECK(snac_mailbox_settle(nullptr) == SNAC_OK && snac_mailbox_quit(nullptr) == SNAC_OK && snac_mailbox_destroy(nullptr) == SNAC_OK);
    CHECK(snac_mailbox_row(nullptr) == nullptr && snac_mailbox_touch(nullptr) == SNAC_ERR_ARG);
    // every fallible call of create() failing in turn (hipHostMalloc, hipStreamCreate): an error, nothing left behind
    const fakehip::Counts base = fakehip::counts();
    d = desc(200);
    for (long k = 0;; ++k) {
        fakehip::fail_nth(k);
        mb = nullptr;
        const int rc = snac_mailbox_create(&d, 0, &mb);
        const bool fired = fakehip::failure_fired();
        fakehip::fail_nth(-1);
        if (!fired) { CHECK(rc == SNAC_OK && mb); CHECK(k == 2); CHECK(snac_mailbox_destroy(mb) == SNAC_OK); break; }
        CHECK(rc == SNAC_ERR_HIP && !mb);
        const fakehip::Counts now = fakehip::counts();
        CHECK(now.mallocs == base.mallocs && now.streams == base.streams);
    }
    const fakehip::Counts end = fakehip::counts();
    CHECK(end.mallocs == base.mallocs && end.streams == base.streams);
}

static void case_one_env_steps_idles_and_comes_back() {
    reset_counts();
    snac_env_desc d = desc(1);
    snac_state st;
    std::memset(&st, 0, sizeof(st));
    snac_mailbox* mb = nullptr;
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK);
    CHECK(snac_mailbox_step(mb, &d, nullptr, 1, 1) == SNAC_ERR_ARG);
    uint32_t stats[8];
    for (int i = 0; i < 20000; ++i) {
        const int act = i % 5, k = 1 + i % 3;
        if (snac_mailbox_step(mb, &d, &st, act, k) != SNAC_OK) { CHECK(!"step"); break; }
        const double* row = snac_mailbox_row(mb);
        if (!(row[0] == (double)(i + 1) && row[1] == (double)act && row[2] == (double)k)) { CHECK(!"row"); break; }
    }
    CHECK(snac_mailbox_step(mb, &d, &st, 77, 9) == SNAC_OK);        // an invalid action steps as -1, the step size is clamped
    CHECK(snac_mailbox_row(mb)[1] == -1.0 && snac_mailbox_row(mb)[2] == 3.0);
    CHECK(snac_mailbox_settle(mb) == SNAC_OK);
    CHECK(snac_mailbox_stats(mb, stats) == SNAC_OK && stats[0] >= 1 && stats[1] == 20001 && stats[3] == 300);
    const int launches0 = (int)stats[0];
    std::this_thread::sleep_for(std::chrono::milliseconds(30));     // 100 idle times: the wave has left by itself
    CHECK(snac_mailbox_stats(mb, stats) == SNAC_OK && stats[2] == 0);
    CHECK(snac_mailbox_step(mb, &d, &st, 2, 2) == SNAC_OK && snac_mailbox_row(mb)[0] == 20002.0);
    CHECK(snac_mailbox_stats(mb, stats) == SNAC_OK && (int)stats[0] == launches0 + 1);
    // generations: touch() makes the wave reload the records before its next step
    const int r0 = g_reloads.load();
    CHECK(snac_mailbox_touch(mb) == SNAC_OK && snac_mailbox_step(mb, &d, &st, 0, 1) == SNAC_OK && g_reloads.load() == r0 + 1);
    CHECK(snac_mailbox_step(mb, &d, &st, 0, 1) == SNAC_OK && g_reloads.load() == r0 + 1);
    // another batch size than the mailbox's is refused (at once: nothing is posted)
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    snac_env_desc d2 = desc(2);
    CHECK(snac_mailbox_step(mb, &d2, &st, 0, 1) == SNAC_ERR_ARG);
    CHECK(g_env_steps[0].load() == 20004);                           // ... and nobody steps it later
    CHECK(snac_mailbox_step(mb, &d, &st, 0, 1) == SNAC_OK && snac_mailbox_row(mb)[0] == 20005.0);
    CHECK(snac_mailbox_quit(mb) == SNAC_OK && snac_mailbox_quit(mb) == SNAC_OK);
    CHECK(snac_mailbox_step(mb, &d, &st, 0, 1) == SNAC_OK && snac_mailbox_row(mb)[0] == 20006.0);   // usable after quit: a new wave
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);                      // with the wave resident
}

static void case_four_waves() {
    for (int n : {2, 64, 65, 130, 256}) {
        reset_counts();
        snac_env_desc d = desc(n);
        snac_state st;
        std::memset(&st, 0, sizeof(st));
        snac_mailbox* mb = nullptr;
        CHECK(snac_mailbox_create(&d, 200, &mb) == SNAC_OK);
        std::vector<int8_t> a(n), k(n);
        for (int t = 0; t < 3000; ++t) {
            for (int e = 0; e < n; ++e) { a[e] = (int8_t)((t + e) % 5); k[e] = (int8_t)(1 + (t + 2 * e) % 3); }
            if (snac_mailbox_step_n(mb, &d, &st, a.data(), k.data()) != SNAC_OK) { CHECK(!"step_n"); break; }
            bool ok = true;
            for (int e = 0; e < n; ++e) {
                const double* row = snac_mailbox_row(mb) + (size_t)e * LD;
                ok = ok && row[0] == (double)(t + 1) && row[1] == (double)a[e] && row[2] == (double)k[e] && row[3] == (double)e;
                ok = ok && snac_mailbox_reward(mb)[e] == (float)a[e] && snac_mailbox_done(mb)[e] == (uint8_t)((t + 1) & 1);
            }
            if (!ok) { CHECK(!"rows of a batch"); break; }
            if (t % 500 == 499) std::this_thread::sleep_for(std::chrono::milliseconds(3));   // the waves leave in between, each by its own clock
        }
        CHECK(snac_mailbox_settle(mb) == SNAC_OK);
        for (int e = 0; e < n; ++e) CHECK(g_env_steps[e].load() == 3000);
        CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
    }
}

static void case_a_queued_launch_is_waited_for() {
    reset_counts();
    snac_env_desc d = desc(1);
    snac_state st;
    std::memset(&st, 0, sizeof(st));
    snac_mailbox* mb = nullptr;
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK);            // limit: the default 120 s
    g_hold_ms = 2600;                                                // past the first look at the stream (2 s)
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(snac_mailbox_step(mb, &d, &st, 3, 2) == SNAC_OK);
    CHECK(ms_since(t0) >= 2500.0 && snac_mailbox_row(mb)[0] == 1.0 && g_env_steps[0].load() == 1);
    CHECK(snac_mailbox_step(mb, &d, &st, 3, 2) == SNAC_OK && snac_mailbox_row(mb)[0] == 2.0);
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
}

static void case_the_limit_withdraws_the_command() {
    reset_counts();
    setenv("SNAC_MAILBOX_TIMEOUT_S", "0.3", 1);
    snac_env_desc d = desc(70);                                      // two waves
    snac_state st;
    std::memset(&st, 0, sizeof(st));
    snac_mailbox* mb = nullptr;
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK);
    std::vector<int8_t> a(70, 1), k(70, 1);
    CHECK(snac_mailbox_step_n(mb, &d, &st, a.data(), k.data()) == SNAC_OK);
    std::this_thread::sleep_for(std::chrono::milliseconds(20));     // both waves gone
    g_hold_ms = 1200;                                                // the waves' launch will sit in a queue for 1.2 s
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = snac_mailbox_step_n(mb, &d, &st, a.data(), k.data());
    const double waited = ms_since(t0);
    CHECK(rc == SNAC_ERR_HIP && std::strstr(snac_last_error(), "withdrawn") != nullptr);
    CHECK(waited >= 290.0 && waited < 1100.0);
    // the waves start 0.9 s after the caller got its error: they find the QUIT and step nothing
    std::this_thread::sleep_for(std::chrono::milliseconds(1300));
    for (int e = 0; e < 70; ++e) CHECK(g_env_steps[e].load() == 1);
    CHECK(__atomic_load_n(&mb->quit_seq[0], __ATOMIC_ACQUIRE) == mb->req_seq && __atomic_load_n(&mb->quit_seq[1], __ATOMIC_ACQUIRE) == mb->req_seq);
    // the mailbox goes on working: the next command is served by fresh waves
    CHECK(snac_mailbox_step_n(mb, &d, &st, a.data(), k.data()) == SNAC_OK);
    for (int e = 0; e < 70; ++e) CHECK(g_env_steps[e].load() == 2);
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
    // a step that is being served while it is withdrawn counts as served
    reset_counts();
    setenv("SNAC_MAILBOX_TIMEOUT_S", "0.01", 1);
    d = desc(1);
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK);
    g_step_ms = 25;                                                  // the "device" takes 25 ms over the step: past the limit, inside the grace
    CHECK(snac_mailbox_step(mb, &d, &st, 2, 1) == SNAC_OK && snac_mailbox_row(mb)[0] == 1.0 && g_env_steps[0].load() == 1);
    g_step_ms = 0;
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
    // a launch cut in two by the limit: wave 0 serves the step, wave 1 starts after the limit -- the error says whose envs stepped
    reset_counts();
    setenv("SNAC_MAILBOX_TIMEOUT_S", "0.2", 1);
    d = desc(100);
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK);
    {
        std::vector<int8_t> a2(100, 2), k2(100, 1);
        g_wave_late_ms[1] = 700;
        const int rc2 = snac_mailbox_step_n(mb, &d, &st, a2.data(), k2.data());
        CHECK(rc2 == SNAC_ERR_HIP && std::strstr(snac_last_error(), "mask 0x1") != nullptr);
        std::this_thread::sleep_for(std::chrono::milliseconds(800));   // wave 1 wakes up, finds the QUIT, steps nothing
        for (int e = 0; e < 100; ++e) CHECK(g_env_steps[e].load() == (e < 64 ? 1 : 0));
        g_wave_late_ms[1] = 0;
        CHECK(snac_mailbox_step_n(mb, &d, &st, a2.data(), k2.data()) == SNAC_OK);
        for (int e = 0; e < 100; ++e) CHECK(g_env_steps[e].load() == (e < 64 ? 2 : 1));
        snac_env_desc d3 = desc(99);                                 // another batch size: refused before its arrays are read
        CHECK(snac_mailbox_step_n(mb, &d3, &st, a2.data(), k2.data()) == SNAC_ERR_ARG);
        for (int e = 0; e < 100; ++e) CHECK(g_env_steps[e].load() == (e < 64 ? 2 : 1));
    }
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
    d = desc(1);
    reset_counts();
    // a launch that fails: the error comes back, the command is withdrawn, the next step works
    unsetenv("SNAC_MAILBOX_TIMEOUT_S");
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK);
    g_launch_fail = 1;
    CHECK(snac_mailbox_step(mb, &d, &st, 2, 1) == SNAC_ERR_HIP && std::strstr(snac_last_error(), "injected") != nullptr);
    CHECK(snac_mailbox_step(mb, &d, &st, 2, 1) == SNAC_OK && g_env_steps[0].load() == 1);
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
}

static void case_the_mailbox_keeps_its_device() {
    reset_counts();
    fakehip::set_devices(2);
    CHECK(hipSetDevice(1) == hipSuccess);
    snac_env_desc d = desc(1);
    snac_state st;
    std::memset(&st, 0, sizeof(st));
    snac_mailbox* mb = nullptr;
    CHECK(snac_mailbox_create(&d, 300, &mb) == SNAC_OK && mb->device == 1);
    CHECK(hipSetDevice(0) == hipSuccess);                            // another device is current when the first step arms the wave
    CHECK(snac_mailbox_step(mb, &d, &st, 1, 1) == SNAC_OK && g_launch_device.load() == 1);
    int cur = -1;
    CHECK(hipGetDevice(&cur) == hipSuccess && cur == 0);             // ... and is current again afterwards
    CHECK(snac_mailbox_destroy(mb) == SNAC_OK);
    CHECK(hipGetDevice(&cur) == hipSuccess && cur == 0);
    fakehip::set_devices(1);
}

int main() {
    const fakehip::Counts base = fakehip::counts();
    case_arguments_and_allocation_failures();
    case_one_env_steps_idles_and_comes_back();
    case_four_waves();
    case_a_queued_launch_is_waited_for();
    case_the_limit_withdraws_the_command();
    case_the_mailbox_keeps_its_device();
    const fakehip::Counts end = fakehip::counts();
    CHECK(end.mallocs == base.mallocs && end.streams == base.streams && end.events == base.events);
    std::printf(g_failed ? "mailbox_host_test: %d check(s) FAILED\n" : "mailbox_host_test: all checks passed\n", g_failed);
    return g_failed ? 1 : 0;
}
