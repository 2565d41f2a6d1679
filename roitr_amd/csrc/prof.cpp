// Optional HIP-event instrumentation (used by bench.py for the `roofline` object): when enabled, the launch
// wrappers bracket their dominant kernel with an event pair on the launch stream and account its
// algorithmic bytes.  Disabled by default: zero events, zero overhead on the product path.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "prof.h"
#include "roitr_engine.h"

namespace {
// pin >= 0: a launch whose work is only known on the device (a batch list with a device-side live length): bytes / aux are PER UNIT,
// the unit count arrives in g_pin[pin] by an async copy queued in front of the launch, and the open phases (phase_mask) get their
// share when the bracket is folded instead of when it is opened
struct Rec { int cls; hipEvent_t a, b; double bytes, aux; int pin = -1; unsigned phase_mask = 0; };
constexpr int PIN_SLOTS = 4096;
int* g_pin = nullptr;
int g_pin_next = 0;
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_open;   // begun, not ended (one per class at a time)
std::vector<Rec> g_done;
std::vector<hipEvent_t> g_pool;
double g_ms[ROITR_PROF_CLASSES], g_bytes[ROITR_PROF_CLASSES], g_aux[ROITR_PROF_CLASSES];
long g_launches[ROITR_PROF_CLASSES];
double g_next_bytes[ROITR_PROF_CLASSES];

bool is_mfma(int cls) { return cls == ROITR_PROF_GEMM || cls == ROITR_PROF_GEMM_HBM || cls == ROITR_PROF_GEO_EMBED; }

void fold(Rec& r)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
        double bytes = r.bytes, aux = r.aux;
        if (r.pin >= 0) {
            const double units = (double)g_pin[r.pin];
            bytes *= units; aux *= units;
            for (int ph = ROITR_PROF_PH_GEOM; ph <= ROITR_PROF_PH_FORWARD; ++ph)
                if (r.phase_mask & (1u << ph)) {
                    if (is_mfma(r.cls)) { g_bytes[ph] += bytes; g_aux[ph] += aux; } else g_aux[ph] += bytes;
                }
        }
        g_ms[r.cls] += ms; g_bytes[r.cls] += bytes; g_aux[r.cls] += aux; g_launches[r.cls] += 1;
    }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
}

hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// fold finished brackets from the front of the queue back into the totals and the event pool (non-blocking), so a long
// timed region re-uses a few thousand events instead of creating two per bracketed launch
void recycle()
{
    size_t n = 0;
    while (n < g_done.size() && hipEventQuery(g_done[n].b) == hipSuccess) {
        fold(g_done[n]);
        ++n;
    }
    if (n) g_done.erase(g_done.begin(), g_done.begin() + n);
}

void drain()
{
    for (auto& r : g_done) {
        (void)hipEventSynchronize(r.b);
        fold(r);
    }
    g_done.clear();
}
}  // namespace

extern "C" int roitr_prof_is_enabled(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_on ? 1 : 0;
}

extern "C" void roitr_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
}

extern "C" void roitr_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    while (g_pool.size() < 4096) {   // events are created here, not inside the region that is about to be timed
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_pool.push_back(e);
    }
    for (int i = 0; i < ROITR_PROF_CLASSES; ++i) { g_ms[i] = 0; g_bytes[i] = 0; g_aux[i] = 0; g_launches[i] = 0; }
}

extern "C" int roitr_prof_read(int cls, double* ms, long* launches, double* bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (cls < 0 || cls >= ROITR_PROF_CLASSES) return 1;
    drain();
    *ms = g_ms[cls]; *launches = g_launches[cls]; *bytes = g_bytes[cls];
    return 0;
}

// the second accumulator of a class: algorithmic HBM bytes of the MFMA classes (whose "bytes" carry FLOPs); for the engine
// phases the algorithmic HBM bytes of every instrumented launch inside the phase
extern "C" int roitr_prof_read_aux(int cls, double* aux)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (cls < 0 || cls >= ROITR_PROF_CLASSES) return 1;
    drain();
    *aux = g_aux[cls];
    return 0;
}

// bytes for the next begin() of `cls` that passes a negative byte count (the wrapper does not know the sizes)
extern "C" void roitr_prof_next_bytes(int cls, double bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (cls >= 0 && cls < ROITR_PROF_CLASSES) g_next_bytes[cls] = bytes;
}

void roitr_prof_begin(int cls, double bytes, hipStream_t st) { roitr_prof_begin2(cls, bytes, 0.0, st); }

static void begin_impl(int cls, double bytes, double aux, const int* dev_units, hipStream_t st);
void roitr_prof_begin2(int cls, double bytes, double aux, hipStream_t st) { begin_impl(cls, bytes, aux, nullptr, st); }
// bytes / aux PER UNIT; the number of units is the device int *dev_units at the time the launch runs (fetched by an async copy on `st`)
void roitr_prof_begin_live(int cls, double bytes_per_unit, double aux_per_unit, const int* dev_units, hipStream_t st)
{
    begin_impl(cls, bytes_per_unit, aux_per_unit, dev_units, st);
}

static void begin_impl(int cls, double bytes, double aux, const int* dev_units, hipStream_t st)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pool.size() < 2 && g_done.size() > 256) recycle();
    Rec r; r.cls = cls; r.bytes = bytes >= 0.0 ? bytes : g_next_bytes[cls]; r.aux = aux; r.a = get_event(); r.b = get_event();
    if (bytes < 0.0) g_next_bytes[cls] = 0.0;   // consumed: a later launch of the class without its own figure counts 0
    if (dev_units) {
        if (!g_pin && hipHostMalloc((void**)&g_pin, sizeof(int) * PIN_SLOTS, hipHostMallocDefault) != hipSuccess) g_pin = nullptr;
        if (g_pin) {
            r.pin = g_pin_next; g_pin_next = (g_pin_next + 1) % PIN_SLOTS;
            g_pin[r.pin] = 0;
            (void)hipMemcpyAsync(&g_pin[r.pin], dev_units, sizeof(int), hipMemcpyDeviceToHost, st);
        }
    }
    // Work issued inside an open engine phase is also booked on the phase: the "bytes" of a phase class are the FLOPs of its
    // GEMM / geo_embed launches (bench.py prices the global-transformer phase against the MFMA peak with them), its "aux" the
    // algorithmic HBM bytes of every instrumented launch inside it (MFMA classes carry them in aux, the others in bytes)
    const bool mfma = is_mfma(cls);
    const bool phase = cls >= ROITR_PROF_PH_GEOM && cls <= ROITR_PROF_PH_FORWARD;
    if (!phase && cls != ROITR_PROF_GEO_ALGO)
        for (auto& o : g_open)
            if (o.cls >= ROITR_PROF_PH_GEOM && o.cls <= ROITR_PROF_PH_FORWARD) {
                if (r.pin >= 0) r.phase_mask |= 1u << o.cls;   // priced when the unit count has arrived (fold)
                else if (mfma) { o.bytes += r.bytes; o.aux += r.aux; }
                else if (cls == ROITR_PROF_LOCAL_BLOCK) { o.bytes += r.aux; o.aux += r.bytes; }   // HBM bytes in `bytes`, FLOPs in `aux`
                else o.aux += r.bytes;
            }
    (void)hipEventRecord(r.a, st);
    g_open.push_back(r);
}

void roitr_prof_note(int cls, double v)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (cls >= 0 && cls < ROITR_PROF_CLASSES) { g_bytes[cls] += v; g_launches[cls] += 1; }
}

void roitr_prof_end(int cls, hipStream_t st)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_open.size(); i-- > 0;) {
        if (g_open[i].cls == cls) {
            (void)hipEventRecord(g_open[i].b, st);
            g_done.push_back(g_open[i]);
            g_open.erase(g_open.begin() + i);
            return;
        }
    }
}

double roitr_gemm_algorithmic_bytes(const RoitrGemm* g)
{
    const double ea = (g->bf16 & ROITR_BF16_A) ? 2.0 : 4.0, ew = (g->bf16 & ROITR_BF16_W) ? 2.0 : 4.0, ec = (g->bf16 & ROITR_BF16_C) ? 2.0 : 4.0;
    const double M = g->M, N = g->N, K = g->K;
    double per = M * K * ea + N * K * ew + M * N * ec;
    if (g->A2) per += M * K * 4.0;
    if (g->ln_res) per += M * N * 4.0;
    if (g->ln_post) per += M * N * 4.0;
    return per * g->batch;
}

int roitr_gemm_prof_class(const RoitrGemm* g)
{
    const double flops = 2.0 * g->M * g->N * (double)g->K * g->batch;
    const double balance = (g->bf16 & ROITR_BF16_W) ? 2500.0 / 8.0 : 157.3 / 8.0;   // peak FLOP/s over peak HBM bytes/s (MI355X_MICROARCH.md)
    return flops < balance * roitr_gemm_algorithmic_bytes(g) ? ROITR_PROF_GEMM_HBM : ROITR_PROF_GEMM;
}
