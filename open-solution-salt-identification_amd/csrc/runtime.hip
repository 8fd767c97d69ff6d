// runtime.hip — error text, device info, the static-program executor and hipGraph capture/replay.
//
// The reference drives its network through Python autograd, one torch operator at a time
// (models.py:105-136).  Here a step is a flat, static list of operator descriptors built once by the
// host; running it is ONE call that enqueues every kernel on a HIP stream, and the same list can be
// captured into a hipGraph so that a launch-bound step costs one graph launch.
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include "common.h"

static thread_local char g_err[512] = "";

void salt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* salt_last_error(void) { return g_err; }

extern "C" int salt_abi_version(void) { return 24; }

extern "C" int salt_device_info(int* cu_count, int* lds_bytes, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) SALT_FAIL((int)e, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) SALT_FAIL((int)e, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch_name && arch_name_len > 0) { strncpy(arch_name, prop.gcnArchName, arch_name_len - 1); arch_name[arch_name_len - 1] = 0; }
    return SALT_OK;
}

extern "C" int salt_program_run_range(const salt_program_entry* e, int begin, int end, void* stream) {
    if (!e || begin < 0 || end < begin) SALT_FAIL(SALT_E_BADARG, "program: bad range");
    for (int i = begin; i < end; ++i) {
        const int rc = e[i].fn(e[i].args, stream);
        if (rc) {
            char prev[400];
            strncpy(prev, g_err, sizeof(prev) - 1); prev[sizeof(prev) - 1] = 0;
            salt_set_error("program entry %d failed (%d): %s", i, rc, prev);
            return rc;
        }
    }
    return SALT_OK;
}

// Same as salt_program_run_range but brackets every entry with HIP events recorded on `stream` (the stream the
// kernels are launched on) and returns the elapsed milliseconds per entry.  Used by bench.py for the live roofline.
extern "C" int salt_program_run_timed(const salt_program_entry* e, int begin, int end, void* stream, float* ms_out) {
    if (!e || begin < 0 || end < begin || !ms_out) SALT_FAIL(SALT_E_BADARG, "program: bad range");
    const int n = end - begin;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t* ev = new hipEvent_t[n + 1];
    for (int i = 0; i <= n; ++i) if (hipEventCreate(&ev[i]) != hipSuccess) { delete[] ev; SALT_FAIL(SALT_E_BADARG, "hipEventCreate failed"); }
    int rc = SALT_OK;
    (void)hipEventRecord(ev[0], st);
    for (int i = 0; i < n && rc == SALT_OK; ++i) {
        rc = e[begin + i].fn(e[begin + i].args, stream);
        (void)hipEventRecord(ev[i + 1], st);
    }
    (void)hipStreamSynchronize(st);
    if (rc == SALT_OK) for (int i = 0; i < n; ++i) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); ms_out[i] = ms; }
    for (int i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}

namespace {
// The fork / join events only order two streams of ONE device: a device-scope release is all they need.  The default event does a
// system-scope fence (L2 write-back + invalidate for host visibility) when it transitions to recorded.  SALT_EVENT_FLAGS: 0 default
// events, 1 hipEventDisableSystemFence, 2 hipEventReleaseToDevice.
unsigned fork_event_flags() {
    static const int mode = getenv("SALT_EVENT_FLAGS") ? atoi(getenv("SALT_EVENT_FLAGS")) : 0;
    return mode == 1 ? hipEventDisableSystemFence : (mode == 2 ? hipEventReleaseToDevice : 0u);
}
struct EventPool {
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t ax[4] = {nullptr, nullptr, nullptr, nullptr};   // auxiliary stream: [0] side -> aux hand-over, [1], [2] "slab free again" ring, [3] main -> aux
    int ensure() {
        for (int i = 0; i < 2; ++i)
            if (!ev[i] && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | fork_event_flags()) != hipSuccess) return -1;
        for (int i = 0; i < 4; ++i)
            if (!ax[i] && hipEventCreateWithFlags(&ax[i], hipEventDisableTiming | fork_event_flags()) != hipSuccess) return -1;
        return 0;
    }
};
thread_local EventPool g_events;
thread_local hipStream_t g_aux_stream = nullptr;   // salt_set_aux_stream: where stream-tag-4 entries run (NULL: on the side stream)
thread_local hipEvent_t g_fork_event = nullptr;
const bool g_fork_handoff_env = !getenv("SALT_NO_FORK_HANDOFF");
thread_local bool g_capturing = false;         // hipExtLaunchKernelGGL stop events are not capturable: plain event forks while a graph records
#define g_fork_handoff (g_fork_handoff_env && !g_capturing)
}  // namespace

hipEvent_t salt_take_fork_event() { hipEvent_t e = g_fork_event; g_fork_event = nullptr; return e; }

extern "C" int salt_set_aux_stream(void* stream) { g_aux_stream = (hipStream_t)stream; return SALT_OK; }

extern "C" int salt_program_run_streams(const salt_program_entry* e, int begin, int end, void* main_stream, void* side_stream) {
    return salt_program_run_streams_ex(e, begin, end, main_stream, side_stream, 1);
}

extern "C" int salt_program_run_streams_ex(const salt_program_entry* e, int begin, int end, void* main_stream, void* side_stream, int join_at_end) {
    return salt_program_run_streams_marks(e, begin, end, main_stream, side_stream, join_at_end, nullptr, 0, nullptr, nullptr);
}

extern "C" int salt_event_create(void** out) {
    if (!out) SALT_FAIL(SALT_E_BADARG, "event_create: null");
    hipEvent_t ev = nullptr;
    const hipError_t err = hipEventCreateWithFlags(&ev, hipEventDisableTiming | fork_event_flags());
    if (err != hipSuccess) SALT_FAIL((int)err, "hipEventCreate: %s", hipGetErrorString(err));
    *out = ev;
    return SALT_OK;
}
extern "C" int salt_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); return SALT_OK; }
extern "C" int salt_stream_wait_event(void* stream, void* ev) {
    if (!ev) SALT_FAIL(SALT_E_BADARG, "stream_wait_event: null event");
    const hipError_t err = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
    if (err != hipSuccess) SALT_FAIL((int)err, "hipStreamWaitEvent: %s", hipGetErrorString(err));
    return SALT_OK;
}

extern "C" int salt_program_run_streams_marks(const salt_program_entry* e, int begin, int end, void* main_stream, void* side_stream, int join_at_end,
                                              const int* marks, int nmarks, void* const* ev_main, void* const* ev_side) {
    if (nmarks < 0 || (nmarks > 0 && (!marks || !ev_main || !ev_side))) SALT_FAIL(SALT_E_BADARG, "program: marks");
    if (!side_stream || side_stream == main_stream) {
        if (nmarks) SALT_FAIL(SALT_E_BADARG, "program: marks need the two-stream executor");
        return salt_program_run_range(e, begin, end, main_stream);
    }
    if (!e || begin < 0 || end < begin) SALT_FAIL(SALT_E_BADARG, "program: bad range");
    if (g_events.ensure()) SALT_FAIL(SALT_E_BADARG, "hipEventCreate failed");
    hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
    static const bool one_stream = getenv("SALT_ONE_STREAM") != nullptr;      // A/B: every entry on the main stream, in program order
    if (one_stream) {
        (void)hipEventRecord(g_events.ev[1], ss);                 // whatever the caller enqueued on the side stream before (weight packs)
        (void)hipStreamWaitEvent(ms, g_events.ev[1], 0);
        int b0 = begin;
        for (int m = 0; m < nmarks; ++m) {
            const int pos = marks[m] < b0 ? b0 : (marks[m] > end ? end : marks[m]);
            const int rc = salt_program_run_range(e, b0, pos, main_stream);
            if (rc) return rc;
            (void)hipEventRecord((hipEvent_t)ev_main[m], ms);
            (void)hipEventRecord((hipEvent_t)ev_side[m], ss);
            b0 = pos;
        }
        return salt_program_run_range(e, b0, end, main_stream);
    }
    bool main_dirty = true, side_used = false;       // main_dirty: main has work the side stream has not been ordered after
    // Fork coalescing (SALT_FORK_EVERY = K > 1): side-stream entries are held back until K groups of them are pending, then issued
    // behind ONE fork.  A held entry only ever runs later than its program position, so its inputs are complete; every cross-queue
    // dependency costs the main queue ~10 us of dispatch stall (rocprofv3 timeline), a held entry costs the side stream its head start.
    static const int fork_eager = getenv("SALT_FORK_EVERY") ? atoi(getenv("SALT_FORK_EVERY")) : 1;
    static const int fork_graph = getenv("SALT_FORK_EVERY_GRAPH") ? atoi(getenv("SALT_FORK_EVERY_GRAPH")) : 8;
    const int fork_every = g_capturing ? fork_graph : fork_eager;   // a cross-stream edge of a replayed graph stalls the main branch ~10 us, a live event does not
    // SALT_FORK_PLAN="a,b,c": group counts of the first flushes of a range (then fork_every): A/B of the flush pattern
    static int plan[16], nplan = -1;
    if (nplan < 0) {
        nplan = 0;
        if (const char* e = getenv("SALT_FORK_PLAN")) {
            while (*e && nplan < 16) { plan[nplan++] = atoi(e); while (*e && *e != ',') ++e; if (*e == ',') ++e; }
        }
    }
    int nflush = 0;
    int pending[64], npending = 0, groups = 0;
    bool prev_side = false;
    auto flush = [&]() -> int {
        if (!npending) return 0;
        if (main_dirty) {
            (void)hipEventRecord(g_events.ev[0], ms);
            (void)hipStreamWaitEvent(ss, g_events.ev[0], 0);
            main_dirty = false;
        }
        for (int k = 0; k < npending; ++k) {
            const int j = pending[k];
            const int rc = e[j].fn(e[j].args, side_stream);
            if (rc) { char prev[400]; strncpy(prev, g_err, sizeof(prev) - 1); prev[sizeof(prev) - 1] = 0; salt_set_error("program entry %d failed (%d): %s", j, rc, prev); return rc; }
        }
        npending = 0; groups = 0; side_used = true; ++nflush;
        return 0;
    };
    // Auxiliary stream (round 6).  Entries tagged 4 - the weight-gradient slab reductions - run on a THIRD stream, each behind the side
    // entry in front of it (its conv_wgrad) and beside the NEXT conv_wgrad, which writes the other of two alternating slab buffers; the
    // side entry that reuses a slab waits for the reduction that read it two reductions ago.  Entries tagged 5 - the optimizer's update
    // of a parameter range whose gradients are final at that position (FusedAdam.backward_program) - run there behind everything both
    // queues were given so far.  Only in the plain eager two-stream run (no marks, no capture, join at the end, a fork per group);
    // otherwise tags 4 and 5 are the side stream (same order guarantees, no concurrency).
    hipStream_t as = g_aux_stream;
    const bool use_aux = as && as != ms && as != ss && !g_capturing && nmarks == 0 && join_at_end && fork_every <= 1;
    int naux = 0;
    bool aux_used = false;
    int mk = 0;
    auto do_marks = [&](int i) -> int {          // every mark at position i: both queues' events, behind everything issued so far
        while (mk < nmarks && marks[mk] <= i) {
            const int rc = flush(); if (rc) return rc;
            (void)hipEventRecord((hipEvent_t)ev_main[mk], ms);
            (void)hipEventRecord((hipEvent_t)ev_side[mk], ss);
            ++mk;
        }
        return 0;
    };
    for (int i = begin; i < end; ++i) {
        if (mk < nmarks) { const int rc = do_marks(i); if (rc) return rc; }
        if (use_aux && e[i].stream == 4) {
            (void)hipEventRecord(g_events.ax[0], ss);            // behind the conv_wgrad that filled the slab
            (void)hipStreamWaitEvent(as, g_events.ax[0], 0);
            const int rc = e[i].fn(e[i].args, as);
            if (rc) { char prev[400]; strncpy(prev, g_err, sizeof(prev) - 1); prev[sizeof(prev) - 1] = 0; salt_set_error("program entry %d failed (%d): %s", i, rc, prev); return rc; }
            (void)hipEventRecord(g_events.ax[1 + (naux & 1)], as);
            ++naux; aux_used = true;
            continue;
        }
        if (use_aux && e[i].stream == 5) {                       // behind BOTH queues (an optimizer update of parameters whose gradients are final here)
            (void)hipEventRecord(g_events.ax[3], ms); (void)hipStreamWaitEvent(as, g_events.ax[3], 0);
            (void)hipEventRecord(g_events.ax[0], ss); (void)hipStreamWaitEvent(as, g_events.ax[0], 0);
            const int rc = e[i].fn(e[i].args, as);
            if (rc) { char prev[400]; strncpy(prev, g_err, sizeof(prev) - 1); prev[sizeof(prev) - 1] = 0; salt_set_error("program entry %d failed (%d): %s", i, rc, prev); return rc; }
            aux_used = true;
            continue;
        }
        const bool side = e[i].stream == 1 || e[i].stream == 4 || e[i].stream == 5;
        if (use_aux && side && i + 1 < end && e[i + 1].stream == 4 && naux >= 2)
            (void)hipStreamWaitEvent(ss, g_events.ax[1 + (naux & 1)], 0);      // the slab this launch writes was read by the reduction two back
        if (fork_every > 1) {
            if (side) {
                if (npending == 64) { const int rc = flush(); if (rc) return rc; }
                pending[npending++] = i; prev_side = true;
                continue;
            }
            if (prev_side) { ++groups; prev_side = false; }
            if (groups >= (nflush < nplan ? plan[nflush] : fork_every) || e[i].stream == 2 || e[i].stream == 3) { const int rc = flush(); if (rc) return rc; }
        }
        if ((e[i].stream == 2 && side_used) || e[i].stream == 3) {   // a main-stream entry that consumes side-stream results:
            (void)hipEventRecord(g_events.ev[1], ss);                 // 2 = produced inside this range, 3 = enqueued on the side
            (void)hipStreamWaitEvent(ms, g_events.ev[1], 0);          // stream before the call (data-gradient weight packs)
            if (aux_used) { (void)hipEventRecord(g_events.ax[0], as); (void)hipStreamWaitEvent(ms, g_events.ax[0], 0); }
            side_used = false;
        }
        if (side && main_dirty) {
            (void)hipEventRecord(g_events.ev[0], ms);
            (void)hipStreamWaitEvent(ss, g_events.ev[0], 0);
            main_dirty = false;
        }
        const bool handoff = g_fork_handoff && fork_every <= 1 && !side && i + 1 < end && e[i + 1].stream == 1;
        if (handoff) g_fork_event = g_events.ev[0];        // the entry may attach it to its last launch as the stop event
        const int rc = e[i].fn(e[i].args, side ? side_stream : main_stream);
        const bool taken = handoff && g_fork_event == nullptr;
        g_fork_event = nullptr;
        if (rc) {
            char prev[400];
            strncpy(prev, g_err, sizeof(prev) - 1); prev[sizeof(prev) - 1] = 0;
            salt_set_error("program entry %d failed (%d): %s", i, rc, prev);
            return rc;
        }
        if (side) side_used = true;
        else if (taken) { (void)hipStreamWaitEvent(ss, g_events.ev[0], 0); main_dirty = false; }
        else main_dirty = true;
    }
    { const int rc = flush(); if (rc) return rc; }
    if (mk < nmarks) { const int rc = do_marks(end); if (rc) return rc; }
    if (side_used && join_at_end) {
        (void)hipEventRecord(g_events.ev[1], ss);
        (void)hipStreamWaitEvent(ms, g_events.ev[1], 0);
    }
    if (aux_used) {                                              // (use_aux implies join_at_end)
        (void)hipEventRecord(g_events.ax[0], as);
        (void)hipStreamWaitEvent(ms, g_events.ax[0], 0);
        (void)hipStreamWaitEvent(ss, g_events.ax[0], 0);          // the next range's first slab writers
    }
    return SALT_OK;
}

extern "C" int salt_program_run(const salt_program_entry* e, int n, void* stream) {
    return salt_program_run_range(e, 0, n, stream);
}

extern "C" int salt_graph_capture(const salt_program_entry* e, int n, void* stream, void** out) {
    if (!out || !stream) SALT_FAIL(SALT_E_BADARG, "graph capture needs a non-default stream");
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) SALT_FAIL((int)err, "hipStreamBeginCapture: %s", hipGetErrorString(err));
    const int rc = salt_program_run(e, n, stream);
    hipGraph_t graph = nullptr;
    err = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (err != hipSuccess) SALT_FAIL((int)err, "hipStreamEndCapture: %s", hipGetErrorString(err));
    hipGraphExec_t exec = nullptr;
    err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (err != hipSuccess) SALT_FAIL((int)err, "hipGraphInstantiate: %s", hipGetErrorString(err));
    *out = (void*)exec;
    return SALT_OK;
}

// Capture of a whole step (several programs, two streams): bracket any sequence of salt_program_run* calls.  The side stream joins
// the capture through the executor's own event forks and must have re-joined `stream` (join_at_end) before salt_graph_end.
extern "C" int salt_graph_begin(void* stream) {
    if (!stream) SALT_FAIL(SALT_E_BADARG, "graph capture needs a non-default stream");
    hipError_t err = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) SALT_FAIL((int)err, "hipStreamBeginCapture: %s", hipGetErrorString(err));
    g_capturing = true;
    return SALT_OK;
}

extern "C" int salt_graph_end(void* stream, void** out) {
    g_capturing = false;
    if (!stream || !out) SALT_FAIL(SALT_E_BADARG, "graph_end: bad args");
    hipGraph_t graph = nullptr;
    hipError_t err = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (err != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); SALT_FAIL((int)err, "hipStreamEndCapture: %s", hipGetErrorString(err)); }
    hipGraphExec_t exec = nullptr;
    err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (err != hipSuccess) SALT_FAIL((int)err, "hipGraphInstantiate: %s", hipGetErrorString(err));
    *out = (void*)exec;
    return SALT_OK;
}

extern "C" int salt_graph_launch(void* exec, void* stream) {
    hipError_t err = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
    if (err != hipSuccess) SALT_FAIL((int)err, "hipGraphLaunch: %s", hipGetErrorString(err));
    return SALT_OK;
}

extern "C" int salt_graph_destroy(void* exec) {
    if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
    return SALT_OK;
}

__global__ void zero_kernel(uint32_t* p, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

extern "C" int salt_zero(const salt_zero_args* a, void* stream) {
    if (!a || !a->p || a->bytes < 0 || (a->bytes & 3)) SALT_FAIL(SALT_E_BADARG, "zero: bad args");
    if (a->bytes == 0) return SALT_OK;
    const int64_t n4 = a->bytes / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)a->p, n4);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

// sizeof() of every struct of saltnet.h in declaration order: the Python host compares them with its
// ctypes mirror at import so that a layout mismatch can never go unnoticed.
extern "C" int salt_abi_struct_sizes(int* out, int n) {
    static const int sizes[] = {
    (int)sizeof(salt_view),
    (int)sizeof(salt_conv_args),
    (int)sizeof(salt_conv_wgrad_args),
    (int)sizeof(salt_wgrad_reduce_args),
    (int)sizeof(salt_wgrad_reduce_batched_args),
    (int)sizeof(salt_pack_conv_weight_args),
    (int)sizeof(salt_pack_batched_args),
    (int)sizeof(salt_conv_first_args),
    (int)sizeof(salt_conv_first_wgrad_args),
    (int)sizeof(salt_s2d_args),
    (int)sizeof(salt_pack_stem_weight_args),
    (int)sizeof(salt_stem_grad_unfold_args),
    (int)sizeof(salt_head1x1_args),
    (int)sizeof(salt_head1x1_bwd_args),
    (int)sizeof(salt_bn_finalize_args),
    (int)sizeof(salt_bn_fold_args),
    (int)sizeof(salt_affine_act_args),
    (int)sizeof(salt_bn_bwd_args),
    (int)sizeof(salt_head_bn_args),
    (int)sizeof(salt_head_bn_bwd_args),
    (int)sizeof(salt_relu_bwd_args),
    (int)sizeof(salt_maxpool2_args),
    (int)sizeof(salt_maxpool2_bwd_args),
    (int)sizeof(salt_avgpool2_args),
    (int)sizeof(salt_bilinear_args),
    (int)sizeof(salt_hyper_rows_args),
    (int)sizeof(salt_hyper_stencil_args),
    (int)sizeof(salt_pad_fold_args),
    (int)sizeof(salt_pad_fold_strip_args),
    (int)sizeof(salt_add_args),
    (int)sizeof(salt_layout_args),
    (int)sizeof(salt_scse_args),
    (int)sizeof(salt_scse_bwd_args),
    (int)sizeof(salt_lovasz_args),
    (int)sizeof(salt_bce_dice_args),
    (int)sizeof(salt_adam_args),
    (int)sizeof(salt_adam_pack_args),
    (int)sizeof(salt_adam_tick_args),
    (int)sizeof(salt_zero_args),
    (int)sizeof(salt_tta_mean_args),
    (int)sizeof(salt_flip_args),
    (int)sizeof(salt_preprocess_args),
    (int)sizeof(salt_crop_threshold_args),
    (int)sizeof(salt_iou_sweep_args),
    (int)sizeof(salt_program_entry)};
    const int m = (int)(sizeof(sizes) / sizeof(sizes[0]));
    for (int i = 0; i < n && i < m; ++i) out[i] = sizes[i];
    return m;
}
