// A stand-in for <hip/hip_runtime.h> for HOST-ONLY sanitizer builds of the library's host side (tools/sanitize_host.sh):
// streams are FIFO queues drained by a thread each, "device memory" is ordinary host memory, events complete when the
// stream has executed everything submitted before their record.  Copies are therefore really asynchronous with respect to
// the threads that issue them -- the staging-slot protocol of csrc/lc_upload.hip (a slot is rewritten only after the event
// behind its previous copy has completed) is exercised, not bypassed: a slot rewritten too early is a data race the thread
// sanitizer reports and a wrong byte the driver's comparison reports.  Nothing here is part of the product.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
struct StubStream;
struct StubEvent;
typedef StubStream* hipStream_t;
typedef StubEvent* hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
enum { hipEventDisableTiming = 2 };
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
struct hipDeviceProp_t { char gcnArchName[256]; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipSetDevice(int dev);
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int dev);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t words, const uint32_t* mask);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemset2DAsync(void* dst, size_t pitch, int value, size_t width, size_t height, hipStream_t s);

// test controls (not HIP): the n-th copy submitted from now on fails (0 = never); a delay of up to `us` microseconds
// before every operation a stream thread executes (shakes out orderings)
void stub_fail_copy_after(long n);
void stub_stream_jitter_us(int us);
long stub_copies_submitted();
