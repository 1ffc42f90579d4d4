/* cuda_on_cpu/NvInferRuntime.h -- TEST INFRASTRUCTURE ONLY: the few TensorRT names that the
 * reference's common/helper.h mentions (none of the plugin classes are compiled). */
#ifndef CUDA_ON_CPU_NVINFER_H
#define CUDA_ON_CPU_NVINFER_H
#include <cstdint>
#include <string>
namespace nvinfer1 {
enum class DataType : int32_t { kFLOAT = 0, kHALF = 1, kINT8 = 2, kINT32 = 3, kBOOL = 4 };
class IPluginCreator {
 public:
  virtual void setPluginNamespace(const char *) noexcept = 0;
  virtual const char *getPluginNamespace() const noexcept = 0;
  virtual ~IPluginCreator() = default;
};
}  // namespace nvinfer1
#endif
