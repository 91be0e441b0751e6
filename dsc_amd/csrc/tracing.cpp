// tracing.cpp — dsc_traces_record / dsc_dump_traces / dsc_clear_traces (dsc.h:159-168).
//
// The reference records begin / end events of every operator on the calling thread and dumps them as a
// Perfetto / chrome://tracing JSON array (dsc_tracing.h:287-310, dsc_tracing.cpp:260-279).  Here an operator call
// only ENQUEUES work, so each record carries two things: the host-side begin / end of the call ("ph": "B" / "E", as the
// reference) and the span its kernels occupied on the context's HIP stream, measured with a pair of events and written
// as a complete event ("ph": "X") on a second track named "HIP stream".
#include "dsc_internal.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>

static unsigned long long now_us() {
    return (unsigned long long) std::chrono::duration_cast<std::chrono::microseconds>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const char *dtype_name(dsc_dtype t) {
    static const char *names[] = {"f32", "f64", "c32", "c64"};
    return t < 4 ? names[t] : "?";
}

static int describe(char *dst, int cap, const char *key, const dsc_tensor *t) {
    int n = snprintf(dst, cap, "\"%s\": {\"shape\": [", key);
    for (int i = DSC_MAX_DIMS - t->n_dim; i < DSC_MAX_DIMS && n < cap; ++i)
        n += snprintf(dst + n, cap - n, "%s%d", i == DSC_MAX_DIMS - t->n_dim ? "" : ", ", t->shape[i]);
    if (n < cap) n += snprintf(dst + n, cap - n, "], \"dtype\": \"%s\", \"backend\": \"MI355X\"}", dtype_name(t->dtype));
    return n < cap ? n : cap - 1;
}

static hipEvent_t take_event(dsc_tracer &tr) {
    if (!tr.free_events.empty()) {
        hipEvent_t e = tr.free_events.back();
        tr.free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}

void dsc_trace_begin(dsc_ctx *ctx, const char *name, const char *cat, const dsc_tensor *a, const dsc_tensor *b, int i0, int i1) {
    dsc_tracer &tr = ctx->tracer;
    if (tr.recs.size() >= (size_t) 1 << 16) {              // bounded, as the reference's fixed trace buffer
        DSC_LOG_INFO("trace buffer full (65536 operator calls): recording stopped");
        tr.recording = false;
        return;
    }
    dsc_trace_rec r;
    memset(&r, 0, sizeof(r));
    snprintf(r.name, sizeof(r.name), "%s", name);
    snprintf(r.cat, sizeof(r.cat), "%s", cat);
    int n = 0;
    const int cap = (int) sizeof(r.args);
    if (a != nullptr) n += describe(r.args + n, cap - n, b != nullptr ? "xa" : "x", a);
    if (b != nullptr && n < cap - 2) { n += snprintf(r.args + n, cap - n, ", "); n += describe(r.args + n, cap - n, "xb", b); }
    if ((i0 != 0 || i1 != 0) && n < cap - 2) snprintf(r.args + n, cap - n, "%s\"n\": %d, \"axis\": %d", n ? ", " : "", i0, i1);
    r.ev_b = take_event(tr);
    r.ev_e = take_event(tr);
    r.ts_b = now_us();
    HIP_CHECK(hipEventRecord(r.ev_b, ctx->stream));
    tr.recs.push_back(r);
}

void dsc_trace_end(dsc_ctx *ctx, size_t index) {
    dsc_trace_rec &r = ctx->tracer.recs[index];
    HIP_CHECK(hipEventRecord(r.ev_e, ctx->stream));
    r.ts_e = now_us();
}

void dsc_trace_release(dsc_ctx *ctx) {
    dsc_tracer &tr = ctx->tracer;
    for (dsc_trace_rec &r : tr.recs) { tr.free_events.push_back(r.ev_b); tr.free_events.push_back(r.ev_e); }
    tr.recs.clear();
}

// dsc.h:162-163
extern "C" void dsc_traces_record(dsc_ctx *ctx, bool record) {
    DSC_ASSERT(ctx != nullptr);
    dsc_tracer &tr = ctx->tracer;
    if (record && !tr.based) {                             // time base: one event whose host time is known
        HIP_CHECK(hipEventCreate(&tr.base_ev));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipEventRecord(tr.base_ev, ctx->stream));
        HIP_CHECK(hipEventSynchronize(tr.base_ev));
        tr.base_ts = now_us();
        tr.based = true;
    }
    tr.recording = record;
}

// dsc.h:165-166, dsc_tracing.cpp:260-279
extern "C" void dsc_dump_traces(dsc_ctx *ctx, const char *filename) {
    DSC_ASSERT(ctx != nullptr && filename != nullptr);
    dsc_tracer &tr = ctx->tracer;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    FILE *f = fopen(filename, "wt");
    DSC_ASSERT(f != nullptr);
    const int pid = (int) getpid();
    fprintf(f, "[\n");
    fprintf(f, "\t{\"name\": \"thread_name\", \"ph\": \"M\", \"pid\": %d, \"tid\": 0, \"args\": {\"name\": \"host: operator calls\"}},\n", pid);
    fprintf(f, "\t{\"name\": \"thread_name\", \"ph\": \"M\", \"pid\": %d, \"tid\": 1, \"args\": {\"name\": \"HIP stream (device %d)\"}}", pid, ctx->device);
    for (const dsc_trace_rec &r : tr.recs) {
        float off_ms = 0.f, dur_ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&off_ms, tr.base_ev, r.ev_b));
        HIP_CHECK(hipEventElapsedTime(&dur_ms, r.ev_b, r.ev_e));
        fprintf(f, ",\n\t{\"name\": \"%s\", \"cat\": \"%s\", \"ph\": \"B\", \"ts\": %llu, \"pid\": %d, \"tid\": 0, \"args\": {%s}}", r.name, r.cat,
                r.ts_b, pid, r.args);
        fprintf(f, ",\n\t{\"name\": \"%s\", \"cat\": \"%s\", \"ph\": \"E\", \"ts\": %llu, \"pid\": %d, \"tid\": 0}", r.name, r.cat, r.ts_e, pid);
        fprintf(f, ",\n\t{\"name\": \"%s\", \"cat\": \"%s;gpu\", \"ph\": \"X\", \"ts\": %.3f, \"dur\": %.3f, \"pid\": %d, \"tid\": 1, \"args\": {%s}}", r.name,
                r.cat, (double) tr.base_ts + (double) off_ms * 1e3, (double) dur_ms * 1e3, pid, r.args);
    }
    fprintf(f, "\n]");
    fclose(f);
    DSC_LOG_INFO("exported Perfetto-compatible traces to \"%s\"", filename);
}

// dsc.h:168
extern "C" void dsc_clear_traces(dsc_ctx *ctx) {
    DSC_ASSERT(ctx != nullptr);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    dsc_trace_release(ctx);
}
