// crossloc_amd._dsacstar_native — the compiled, ATen-level binding of the solver, the counterpart of the reference's own
// pybind11 extension (/root/reference/dsacstar/dsacstar.cpp:887-892: forward_rgb, backward_rgb, forward_rgbd, backward_rgbd).
// Same positional signatures, at::Tensor arguments, the errors at::Tensor::accessor<float, N>() raises (c10::Error ->
// RuntimeError) on a wrong rank or dtype.  Everything below the argument handling is the C ABI of include/crossloc_dsac.h
// (libcrossloc_hip.so): CPU tensors take the host entry point (xl_dsac_forward_rgb_host: H2D, kernels, D2H, synchronise - the
// reference's call is blocking as well), GPU tensors the batched device entry points on the current stream.  No HIP header is
// needed here: the current stream comes from torch.cuda.current_stream(), the device is set with a c10::DeviceGuard.
// The module is host-only C++ (g++), built in-tree by crossloc_amd/build.py next to the library; `dsacstar.py` prefers it and
// falls back to the ctypes shim (crossloc_amd/dsacstar.py) when it is not built.  The sampler's image counter lives in the
// Python shim (one counter whichever binding is used).
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>

#include "../../include/crossloc_dsac.h"

namespace py = pybind11;

namespace {

constexpr unsigned long long kSeed = 1305;            // thread_rand.h:101 of the reference
constexpr unsigned kMaxTries = 1000000;               // dsacstar.cpp:48

void check_status(int rc)
{
    if (rc == 0) return;
    std::string msg = std::string("crossloc_hip: ") + xl_status_string(rc);
    if (rc == XL_ERR_HIP) msg += std::string(": ") + xl_last_hip_error();
    throw std::runtime_error(msg + " (status " + std::to_string(rc) + ")");
}

long long take_image_index()
{
    return py::module_::import("crossloc_amd.dsacstar").attr("_take_image_index")().cast<long long>();
}

void *current_stream(const at::Tensor &t)
{
    py::object s = py::module_::import("torch").attr("cuda").attr("current_stream")(py::cast(t.device()));
    return reinterpret_cast<void *>(s.attr("cuda_stream").cast<uintptr_t>());
}

void check_coords(const at::Tensor &t)
{
    (void)t.accessor<float, 4>();                     // rank / dtype errors exactly as the reference raises them (dsacstar.cpp:78)
    TORCH_CHECK(t.size(1) == 3, "sceneCoordinates must be [1,3,H,W], got ", t.sizes());
    TORCH_CHECK(t.size(0) == 1, "forward_rgb supports batch size 1 only (dsacstar_util.h:161); use forward_rgb_batch");
}

// dsacstar_rgb_forward (dsacstar.cpp:63-178)
void forward_rgb(at::Tensor sceneCoordinatesSrc, at::Tensor outPoseSrc, int ransacHypotheses, float inlierThreshold,
                 float focalLength, float ppointX, float ppointY, float inlierAlpha, float maxReproj, int subSampling)
{
    check_coords(sceneCoordinatesSrc);
    TORCH_CHECK(outPoseSrc.dim() == 2 && outPoseSrc.size(0) == 4 && outPoseSrc.size(1) == 4 && outPoseSrc.scalar_type() == at::kFloat,
                "outPose must be a float32 [4,4] tensor");
    const long long image = take_image_index();
    const int Ho = (int)sceneCoordinatesSrc.size(2), Wo = (int)sceneCoordinatesSrc.size(3);
    if (sceneCoordinatesSrc.is_cuda()) {
        at::Tensor dst = (outPoseSrc.is_cuda() && outPoseSrc.is_contiguous()) ? outPoseSrc
                         : at::empty({4, 4}, sceneCoordinatesSrc.options());
        {
            c10::DeviceGuard guard(sceneCoordinatesSrc.device());
            py::gil_scoped_acquire gil;
            void *stream = current_stream(sceneCoordinatesSrc);
            check_status(xl_dsac_forward_rgb_batch(sceneCoordinatesSrc.data_ptr<float>(), sceneCoordinatesSrc.stride(0), sceneCoordinatesSrc.stride(1),
                                                   sceneCoordinatesSrc.stride(2), sceneCoordinatesSrc.stride(3), 1, Ho, Wo, dst.data_ptr<float>(),
                                                   ransacHypotheses, inlierThreshold, focalLength, ppointX, ppointY, inlierAlpha, maxReproj,
                                                   subSampling, nullptr, kSeed, (uint64_t)image, 1, kMaxTries, stream,
                                                   nullptr, nullptr, nullptr, nullptr));
        }
        if (!dst.is_same(outPoseSrc)) outPoseSrc.copy_(dst);          // (a synchronising copy when outPose lives on the host)
        return;
    }
    TORCH_CHECK(!outPoseSrc.is_cuda(), "outPose on the GPU needs sceneCoordinates on the GPU too");
    at::Tensor host = outPoseSrc.is_contiguous() ? outPoseSrc : at::empty({4, 4}, outPoseSrc.options());
    check_status(xl_dsac_forward_rgb_host(sceneCoordinatesSrc.data_ptr<float>(), sceneCoordinatesSrc.stride(1), sceneCoordinatesSrc.stride(2),
                                          sceneCoordinatesSrc.stride(3), Ho, Wo, host.data_ptr<float>(), ransacHypotheses, inlierThreshold,
                                          focalLength, ppointX, ppointY, inlierAlpha, maxReproj, subSampling, kSeed, (uint64_t)image,
                                          kMaxTries, nullptr, nullptr, nullptr, nullptr));
    if (!host.is_same(outPoseSrc)) outPoseSrc.copy_(host);
}

// dsacstar_rgb_backward (dsacstar.cpp:200-483): returns the expected pose loss, accumulates the gradient
double backward_rgb(at::Tensor sceneCoordinatesSrc, at::Tensor outSceneCoordinatesGradSrc, at::Tensor gtPoseSrc, int ransacHypotheses,
                    float inlierThreshold, float focalLength, float ppointX, float ppointY, float wLossRot, float wLossTrans,
                    float softClamp, float inlierAlpha, float maxReproj, int subSampling, int randomSeed)
{
    check_coords(sceneCoordinatesSrc);
    (void)outSceneCoordinatesGradSrc.accessor<float, 4>();
    TORCH_CHECK(outSceneCoordinatesGradSrc.sizes() == sceneCoordinatesSrc.sizes(),
                "outSceneCoordinatesGrad must be float32 with the shape of sceneCoordinates");
    TORCH_CHECK(gtPoseSrc.dim() == 2 && gtPoseSrc.size(0) == 4 && gtPoseSrc.size(1) == 4, "gtPose must be a [4,4] tensor");
    TORCH_CHECK(sceneCoordinatesSrc.is_cuda() == outSceneCoordinatesGradSrc.is_cuda(),
                "sceneCoordinates and outSceneCoordinatesGrad must be on the same device");
    const bool onHost = !sceneCoordinatesSrc.is_cuda();
    // host tensors: there is no CPU fallback - the work runs on the current HIP device and the gradient is copied back
    at::Tensor co = onHost ? sceneCoordinatesSrc.to(at::Device(at::kCUDA)) : sceneCoordinatesSrc;
    at::Tensor gd = onHost ? outSceneCoordinatesGradSrc.to(co.device()).contiguous() : outSceneCoordinatesGradSrc;
    at::Tensor gt = gtPoseSrc.to(co.device(), at::kFloat).reshape({1, 16}).contiguous();
    at::Tensor loss = at::zeros({1}, co.options().dtype(at::kDouble));
    const int Ho = (int)co.size(2), Wo = (int)co.size(3);
    {
        c10::DeviceGuard guard(co.device());
        void *stream = current_stream(co);
        check_status(xl_dsac_backward_rgb_batch(co.data_ptr<float>(), co.stride(0), co.stride(1), co.stride(2), co.stride(3), 1, Ho, Wo,
                                                gd.data_ptr<float>(), gd.stride(0), gd.stride(1), gd.stride(2), gd.stride(3),
                                                gt.data_ptr<float>(), loss.data_ptr<double>(), ransacHypotheses, inlierThreshold,
                                                focalLength, ppointX, ppointY, wLossRot, wLossTrans, softClamp, inlierAlpha, maxReproj,
                                                subSampling, nullptr, (uint64_t)randomSeed, 0, 1, kMaxTries, stream, nullptr));
    }
    if (onHost) outSceneCoordinatesGradSrc.copy_(gd);
    return loss.item<double>();
}

[[noreturn]] void not_implemented(const char *what)
{
    PyErr_SetString(PyExc_NotImplementedError, what);
    throw py::error_already_set();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X-native dsacstar: compiled binding over the C ABI of libcrossloc_hip.so (include/crossloc_dsac.h)";
    m.def("forward_rgb", &forward_rgb, "Performs pose estimation from RGB (forward pass).");          // dsacstar.cpp:888
    m.def("backward_rgb", &backward_rgb, "Performs pose estimation from RGB and calculates the gradients of the pose loss "
                                         "wrt. to the input scene coordinates.");                    // dsacstar.cpp:889
    m.def("forward_rgbd", [](py::args, py::kwargs) {                                                   // dsacstar.cpp:890
        not_implemented("dsacstar.forward_rgbd is not on CrossLoc's path and is not implemented");
    });
    m.def("backward_rgbd", [](py::args, py::kwargs) {                                                  // dsacstar.cpp:891
        not_implemented("dsacstar.backward_rgbd is not on CrossLoc's path and is not implemented");
    });
}
