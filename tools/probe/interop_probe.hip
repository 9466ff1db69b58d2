// Feasibility probe for the HIP half of Vulkan<->HIP interop: a device allocation is exported as a
// POSIX fd (dma-buf, what VK_KHR_external_memory_fd hands out on amdgpu) and imported back through
// hipImportExternalMemory / hipExternalMemoryGetMappedBuffer; the two mappings must alias.  No kernels: copies only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %d (%s)\n", #x, (int)e_, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    CK(hipSetDevice(0));
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t size = ((size_t)(8u << 20) + gran - 1) / gran * gran;
    printf("granularity %zu size %zu\n", gran, size);
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, size, &prop, 0));
    void *a = nullptr;
    CK(hipMemAddressReserve(&a, size, 0, nullptr, 0));
    CK(hipMemMap(a, size, 0, h, 0));
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(a, size, &acc, 1));
    int fd = -1;
    CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
    printf("exported fd %d\n", fd);
    hipExternalMemoryHandleDesc d;
    memset(&d, 0, sizeof(d));
    d.type = hipExternalMemoryHandleTypeOpaqueFd;
    d.handle.fd = fd;
    d.size = size;
    hipExternalMemory_t em;
    CK(hipImportExternalMemory(&em, &d));
    hipExternalMemoryBufferDesc bd;
    memset(&bd, 0, sizeof(bd));
    bd.size = size;
    void *b = nullptr;
    CK(hipExternalMemoryGetMappedBuffer(&b, em, &bd));
    printf("original mapping %p, imported mapping %p\n", a, b);
    const size_t n = size / 4;
    unsigned *hbuf = (unsigned *)malloc(size), *back = (unsigned *)malloc(size);
    for (size_t i = 0; i < n; i++) hbuf[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(a, hbuf, size, hipMemcpyHostToDevice));
    CK(hipMemcpy(back, b, size, hipMemcpyDeviceToHost));
    printf("write through original, read through import: %s\n", memcmp(hbuf, back, size) ? "MISMATCH" : "equal");
    for (size_t i = 0; i < n; i++) hbuf[i] ^= 0x5A5A5A5Au;
    CK(hipMemcpy(b, hbuf, size, hipMemcpyHostToDevice));
    CK(hipMemcpy(back, a, size, hipMemcpyDeviceToHost));
    printf("write through import, read through original: %s\n", memcmp(hbuf, back, size) ? "MISMATCH" : "equal");
    CK(hipFree(b));
    CK(hipDestroyExternalMemory(em));
    CK(hipMemUnmap(a, size));
    CK(hipMemAddressFree(a, size));
    CK(hipMemRelease(h));
    close(fd);
    printf("released\n");
    return 0;
}
