// external_event_probe.hip -- can work on ANOTHER stream be ordered behind a point INSIDE a replaying hipGraph?   (standalone:
// hipcc --offload-arch=gfx950 -O2 -o external_event_probe external_event_probe.hip)
// the data-parallel all-reduce of a gradient bucket should start under the tail of the backward.  A lane's micro-batch is ONE captured
// graph, so the point "the last layers' gradients are final" lies inside it.  torch 2.10+rocm7.0 refuses torch.cuda.Event(external=True) under capture
// ("External events are disallowed in rocm", tools/external_event_probe.py); this probe asks the HIP runtime itself, two ways:
//   A. an event-record NODE: hipEventRecordWithFlags(ev, capturing stream, hipEventRecordExternal) between two spin kernels; after every replay stream B
//      does hipStreamWaitEvent(B, ev) and stamps an event.  Want: B released after the FIRST spin of THIS replay (~T), not at once (a stale record), not after 2T.
//   B. a COUNTER kernel node (atomicAdd on a device word) between the spins, stream B waits with hipStreamWaitValue32(B, word, replay count, GTE).
// Prints one JSON line per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define SOFT(x, what) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"case\": \"%s\", \"unsupported\": \"%s at line %d\"}\n", what, hipGetErrorString(e_), __LINE__); (void)hipGetLastError(); goto next; } } while (0)

__global__ void spin(long ticks, int* sink) {
    const long t0 = __builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < ticks) { }
    if (sink && threadIdx.x == 0) atomicAdd(sink, 0);
}
__global__ void bump(unsigned* word) { if (threadIdx.x == 0) { __threadfence_system(); atomicAdd_system(word, 1u); } }

static float ms(hipEvent_t a, hipEvent_t b) { float t = 0.f; (void)hipEventElapsedTime(&t, a, b); return t; }

int main() {
    const long T = 100000L * 20;          // s_memtime ticks at 100 MHz: 20 ms
    hipStream_t A, B;
    CHECK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t t0, tm, tg;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&tm)); CHECK(hipEventCreate(&tg));
    int* sink; CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(sink, 0, 4));
    spin<<<1, 64, 0, A>>>(1000, sink); CHECK(hipStreamSynchronize(A));

    {   // ---- A: external event-record node
        hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipGraph_t g; hipGraphExec_t ge;
        SOFT(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal), "event_node");
        spin<<<1, 64, 0, A>>>(T, sink);
        {
            hipError_t e = hipEventRecordWithFlags(ev, A, hipEventRecordExternal);
            if (e != hipSuccess) {
                printf("{\"case\": \"event_node\", \"unsupported\": \"hipEventRecordWithFlags(External) under capture: %s\"}\n", hipGetErrorString(e));
                (void)hipStreamEndCapture(A, &g); (void)hipGetLastError(); goto next;
            }
        }
        spin<<<1, 64, 0, A>>>(T, sink);
        SOFT(hipStreamEndCapture(A, &g), "event_node");
        SOFT(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0), "event_node");
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(t0, A));
            CHECK(hipGraphLaunch(ge, A));
            CHECK(hipEventRecord(tg, A));
            SOFT(hipStreamWaitEvent(B, ev, 0), "event_node");
            CHECK(hipEventRecord(tm, B));
            CHECK(hipDeviceSynchronize());
            printf("{\"case\": \"event_node\", \"replay\": %d, \"t_mark_ms\": %.2f, \"t_graph_ms\": %.2f, \"want_mark_ms\": 20, \"want_graph_ms\": 40}\n", rep, ms(t0, tm), ms(t0, tg));
        }
    }
next:
    {   // ---- B: counter kernel node + hipStreamWaitValue32
        unsigned* word; CHECK(hipMalloc(&word, 4)); CHECK(hipMemset(word, 0, 4));
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        spin<<<1, 64, 0, A>>>(T, sink);
        bump<<<1, 64, 0, A>>>(word);
        spin<<<1, 64, 0, A>>>(T, sink);
        CHECK(hipStreamEndCapture(A, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(t0, A));
            CHECK(hipGraphLaunch(ge, A));
            CHECK(hipEventRecord(tg, A));
            hipError_t e = hipStreamWaitValue32(B, word, (unsigned)(rep + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf("{\"case\": \"wait_value\", \"unsupported\": \"hipStreamWaitValue32: %s\"}\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
            CHECK(hipEventRecord(tm, B));
            CHECK(hipDeviceSynchronize());
            printf("{\"case\": \"wait_value\", \"replay\": %d, \"t_mark_ms\": %.2f, \"t_graph_ms\": %.2f, \"want_mark_ms\": 20, \"want_graph_ms\": 40}\n", rep, ms(t0, tm), ms(t0, tg));
        }
        // the same with the wait enqueued BEFORE the launch (the value it waits for does not exist yet: must still release at ~T)
        for (int rep = 5; rep < 8; ++rep) {
            CHECK(hipEventRecord(t0, A));
            CHECK(hipStreamWaitEvent(B, t0, 0));
            hipError_t e = hipStreamWaitValue32(B, word, (unsigned)(rep + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { (void)hipGetLastError(); break; }
            CHECK(hipEventRecord(tm, B));
            CHECK(hipGraphLaunch(ge, A));
            CHECK(hipEventRecord(tg, A));
            CHECK(hipDeviceSynchronize());
            printf("{\"case\": \"wait_value_enqueued_first\", \"replay\": %d, \"t_mark_ms\": %.2f, \"t_graph_ms\": %.2f, \"want_mark_ms\": 20, \"want_graph_ms\": 40}\n", rep, ms(t0, tm), ms(t0, tg));
        }
    }
    return 0;
}
