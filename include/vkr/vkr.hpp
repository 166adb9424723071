// vkr.hpp - header-only C++17 facade over the C ABI (vkr_c.h) that keeps the names and call shapes of
// the reference's vw:: wrappers for everything a caller touches to run a compute pass, so host code
// written like /root/reference/src/main.cpp:88-92,139,156-173,187-196,200-231 ports by changing the
// namespace and replacing loadShader("...spv") with a kernel name.
//
//   vw:: (reference)                                   vkr:: (this header)
//   Instance + Device + vw::g::device  vkcore.hpp:186-218   Device(ordinal)            - explicit, no globals
//   Queue::oneTimeRecordSubmit/hasReadyBuffer  :124-168     Queue                      - HIP stream + events
//   CommandBuffer (= vk::CommandBuffer)        :50-87       CommandBuffer              - records a vkr_cmd list
//   Fence / Semaphore                          :27-48       Fence / Semaphore          - hipEvent / stream order
//   Shader, ShaderIOBinding, loadShader   vkshader.hpp:9-37 Shader(device, stage, loadShader(path) = kernel name, ioBindings, pushSize)
//   PipelineLayout, DescriptorSetLayout, DedicatedDescriptorPool, DescriptorSet
//                                    vkdescriptor.hpp:8-67  same names; bindings numbered per set in
//                                                           declaration order (vkdescriptor.cpp:56-62)
//   ComputePipeline(layout, shader)   vkcompute.hpp:6-9     ComputePipeline(layout, shader)
//   MemoryAllocator, Buffer, Image, ImageView, StagingBuffer
//                                     vkmemory.hpp:25-219   same names; VMA -> hipMalloc / pinned host
//
// Errors surface as std::runtime_error, as in the reference (src/main.cpp:233-242).  Objects are
// move-only RAII and must be destroyed before their Device (declaration order, as in the reference).
// A Device and everything created from it belong to one thread.
#pragma once

#include <array>
#include <cstdint>
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <filesystem>
#include <functional>
#include <initializer_list>
#include <fstream>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../vkr_c.h"

namespace vkr {

inline void check(int32_t status, const char* what) {
  if (status != VKR_OK) throw std::runtime_error(std::string(what) + ": " + vkr_last_error());
}

// ---------------------------------------------------------------- small utilities (vkutils.hpp:15-54,161-174,229-233)
// A non-owning view of consecutive elements, implicitly constructible from whatever a vw:: call site passes
// where the reference takes a vw::ArrayProxy: one element, a C array, std::array, std::vector, a braced list, or
// pointer + count.  Kept as a [first, last) range.
template <typename T>
class ArrayProxy {
 public:
  using value_type = T;
  ArrayProxy() = default;
  ArrayProxy(const T& single) : mFirst{&single}, mLast{&single + 1} {}
  template <size_t N>
  ArrayProxy(const T (&items)[N]) : mFirst{items}, mLast{items + N} {}
  template <size_t N>
  ArrayProxy(const std::array<T, N>& items) : mFirst{items.data()}, mLast{items.data() + N} {}
  ArrayProxy(const std::vector<std::remove_const_t<T>>& items) : mFirst{items.data()}, mLast{items.data() + items.size()} {}
  ArrayProxy(const std::initializer_list<T>& items) : mFirst{items.begin()}, mLast{items.end()} {}
  ArrayProxy(const T* items, uint32_t count) : mFirst{items}, mLast{items + count} {}
  const T* begin() const { return mFirst; }
  const T* end() const { return mLast; }
  const T* data() const { return mFirst; }
  uint32_t size() const { return static_cast<uint32_t>(mLast - mFirst); }
  bool empty() const { return mFirst == mLast; }
  const T& operator[](size_t i) const { return mFirst[i]; }

 private:
  const T* mFirst = nullptr;
  const T* mLast = nullptr;
};

template <typename T>
inline uint32_t size32(const std::vector<T>& v) {
  return static_cast<uint32_t>(v.size());
}
template <typename T>
inline uint64_t byteSize(const T& container) {
  return sizeof(typename T::value_type) * container.size();
}
inline void alignTo(uint64_t& size, uint64_t alignment) { size = (size + alignment - 1) / alignment * alignment; }

// vw::loadShader (vkshader.hpp:18) returns the SPIR-V words of a file.  The kernels here are compiled into the
// library and looked up by name, so what a call site gets back is that name, taken from the file name the
// reference passes: "shaders/deferred.comp.spv" -> "deferred", "shaders/prep_indirect.comp.spv" -> "prep_indirect".
// The file is not opened.
inline std::string loadShader(const std::filesystem::path& path) {
  const std::string file = path.filename().string();
  const std::string name = file.substr(0, file.find('.'));
  if (name.empty()) throw std::runtime_error("vkr::loadShader: '" + path.string() + "' names no kernel");
  return name;
}

// vw::DataFile / vw::ImageFile (vkutils.hpp:192-206): a source of image bytes for StagingBuffer::queueImageCopy.
class DataFile {
 public:
  virtual ~DataFile() = default;
  virtual size_t dataSize() const = 0;
  virtual void loadData(std::byte* dst) const = 0;
};
class Extent;
enum class Format : uint32_t;
class ImageFile : public DataFile {
 public:
  virtual Format getFormat() const;
  virtual Extent getExtent() const = 0;
};

// ---------------------------------------------------------------- enums with the vk:: spellings the caller uses
enum class Format : uint32_t {
  eUndefined = VKR_FORMAT_UNDEFINED,
  eR8G8B8A8Unorm = VKR_FORMAT_R8G8B8A8_UNORM,
  eB8G8R8A8Unorm = VKR_FORMAT_B8G8R8A8_UNORM,
  eR16G16B16A16Sfloat = VKR_FORMAT_R16G16B16A16_SFLOAT,
  eR32Sfloat = VKR_FORMAT_R32_SFLOAT,
  eR32G32B32A32Sfloat = VKR_FORMAT_R32G32B32A32_SFLOAT,
  eD32Sfloat = VKR_FORMAT_D32_SFLOAT
};
enum class DescriptorType { eCombinedImageSampler, eUniformBuffer, eStorageBuffer, eStorageImage };
enum class ShaderStageFlagBits { eCompute };
enum class PipelineBindPoint { eCompute };
enum class ImageLayout { eUndefined, eGeneral, eShaderReadOnlyOptimal, eTransferDstOptimal, ePresentSrcKHR };
enum MemoryUsage { VKR_MEMORY_USAGE_GPU_ONLY = 0, VKR_MEMORY_USAGE_CPU_TO_GPU = 1 };
using BufferUsageFlags = uint32_t;
namespace BufferUse {
constexpr BufferUsageFlags kStagingBuffer = 1, kUniformBuffer = 2, kStorageBuffer = 4, kIndirectBuffer = 8;
}

inline uint32_t texelBytes(Format f) {
  switch (f) {
    case Format::eR16G16B16A16Sfloat: return 8;
    case Format::eR32G32B32A32Sfloat: return 16;
    case Format::eUndefined: return 0;
    default: return 4;
  }
}

class Extent {
 public:
  Extent(uint32_t width, uint32_t height, uint32_t depth = 1) : width{width}, height{height}, depth{depth} {}
  uint32_t width, height, depth;
};
inline Format ImageFile::getFormat() const { return Format::eUndefined; }

// ---------------------------------------------------------------- device / sync
class Queue;

class Device {
 public:
  explicit Device(int32_t ordinal = 0) { check(vkr_context_create(ordinal, &mCtx), "vkr::Device"); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  ~Device();
  vkr_context* getHandle() const { return mCtx; }
  Queue& getPreferredQueue();
  void waitIdle() { check(vkr_queue_wait_idle(mCtx), "Device::waitIdle"); }
  std::string description() const {
    char buf[256];
    check(vkr_context_device_info(mCtx, buf, sizeof(buf), nullptr, nullptr), "Device::description");
    return buf;
  }

 private:
  vkr_context* mCtx = nullptr;
  std::unique_ptr<Queue> mQueue;
};

class Semaphore {};  // submissions of one Device are stream-ordered; kept so call sites need no edits

class Fence {
 public:
  explicit Fence(vkr_fence* handle) : mHandle{handle} {}
  Fence(const Fence&) = delete;
  Fence& operator=(const Fence&) = delete;
  ~Fence() { vkr_fence_destroy(mHandle); }
  bool signaled() const {
    const int32_t s = vkr_fence_status(mHandle);
    if (s == VKR_ERR_NOT_READY) return false;
    check(s, "Fence::signaled");
    return true;
  }
  void reset() {}
  // The reference waits 1 ms and ignores the result (vkcore.hpp:45-47); this one really waits.
  void wait() { check(vkr_fence_wait(mHandle, UINT64_MAX), "Fence::wait"); }
  vkr_fence* getHandle() const { return mHandle; }

 private:
  vkr_fence* mHandle;
};

// ---------------------------------------------------------------- memory (vkmemory.hpp)
class MemoryAllocator {
 public:
  explicit MemoryAllocator(Device& device) : mDevice{device} {}
  Device& device() { return mDevice; }

 private:
  Device& mDevice;
};

class Buffer;
struct DescriptorBufferInfo {
  const Buffer* buffer = nullptr;
  uint64_t offset = 0;
  uint64_t range = 0;
};

class Buffer {
 public:
  Buffer(MemoryAllocator& allocator, ArrayProxy<uint64_t> segmentSizes, BufferUsageFlags usage, MemoryUsage memoryUsage = VKR_MEMORY_USAGE_GPU_ONLY)
      : mUsage{usage} {
    constexpr uint64_t cAlign = 256;  // vkmemory.cpp:23
    uint64_t next = 0;
    for (const auto& v : segmentSizes) {
      mSegmentBase.push_back(next);
      mSegmentSizes.push_back(v);
      next += v;
      alignTo(next, cAlign);
    }
    check(vkr_buffer_create(allocator.device().getHandle(), next, memoryUsage == VKR_MEMORY_USAGE_CPU_TO_GPU ? VKR_BUFFER_HOST_VISIBLE : VKR_BUFFER_DEVICE, &mHandle),
          "vkr::Buffer");
    if (memoryUsage == VKR_MEMORY_USAGE_CPU_TO_GPU) {
      void* p = nullptr;
      check(vkr_buffer_map(mHandle, &p), "vkr::Buffer map");
      mMappedPtr = static_cast<std::byte*>(p);
    }
  }
  Buffer(const Buffer&) = delete;
  Buffer& operator=(const Buffer&) = delete;
  Buffer(Buffer&& o) noexcept { *this = std::move(o); }
  Buffer& operator=(Buffer&& o) noexcept {
    std::swap(mHandle, o.mHandle);
    std::swap(mSegmentBase, o.mSegmentBase);
    std::swap(mSegmentSizes, o.mSegmentSizes);
    std::swap(mMappedPtr, o.mMappedPtr);
    std::swap(mUsage, o.mUsage);
    return *this;
  }
  ~Buffer() { vkr_buffer_destroy(mHandle); }

  template <typename T>
  void copyToMapped(const T& src, size_t segmentIdx = 0, uint64_t segmentOffset = 0) {
    if (mMappedPtr == nullptr) throw std::runtime_error("Buffer::copyToMapped: buffer is not CPU_TO_GPU");
    if (segmentIdx >= mSegmentBase.size() || segmentOffset + byteSize(src) > mSegmentSizes[segmentIdx])
      throw std::runtime_error("Buffer::copyToMapped: out of range");
    std::memcpy(mMappedPtr + mSegmentBase[segmentIdx] + segmentOffset, src.data(), byteSize(src));
  }
  DescriptorBufferInfo getSegmentDesc(size_t idx) const { return {this, mSegmentBase[idx], mSegmentSizes[idx]}; }
  vkr_buffer* getHandle() const { return mHandle; }
  void* devicePtr() const {
    void* p = nullptr;
    check(vkr_buffer_device_ptr(mHandle, &p), "Buffer::devicePtr");
    return p;
  }
  void upload(const void* src, uint64_t bytes, uint64_t offset = 0) { check(vkr_buffer_upload(mHandle, offset, src, bytes), "Buffer::upload"); }
  void download(void* dst, uint64_t bytes, uint64_t offset = 0) const { check(vkr_buffer_download(mHandle, offset, dst, bytes), "Buffer::download"); }

 protected:
  vkr_buffer* mHandle = nullptr;
  std::vector<uint64_t> mSegmentBase, mSegmentSizes;
  std::byte* mMappedPtr = nullptr;
  BufferUsageFlags mUsage = 0;
};

class Image;
class CommandBuffer;
class ImageView {
 public:
  explicit ImageView(const Image* image) : mImage{image} {}
  const Image* image() const { return mImage; }

 private:
  const Image* mImage;
};

// A linear, row-major 2D image; rows padded to whole 4-texel groups and 16 bytes.
class Image {
 public:
  Image(MemoryAllocator& allocator, Format format, Extent extent, uint32_t /*usage*/ = 0)
      : mFormat{format}, mExtent{extent}, mPitch{rowPitch(format, extent.width)},
        mStorage{allocator, uint64_t(mPitch) * extent.height, BufferUse::kStorageBuffer} {}
  static uint32_t rowPitch(Format f, uint32_t width) { return ((((width + 3u) / 4u) * 4u * texelBytes(f)) + 15u) & ~15u; }
  ImageView createView() const { return ImageView{this}; }
  // vw::Image::transitionLayout (vkmemory.hpp:87-95), both forms, as main.cpp:221,227 call them inside the record
  // lambda.  Linear images in HIP memory have no layouts and submissions of a Device are stream-ordered, so no
  // barrier is recorded; what is kept is the check the validation layer would make: the command buffer must be
  // recording and oldLayout must be eUndefined or the layout the image was last moved to.
  static void transitionLayout(CommandBuffer& cmdBuffer, const Image& image, ImageLayout oldLayout, ImageLayout newLayout) {
    image.transitionLayout(cmdBuffer, oldLayout, newLayout);
  }
  inline void transitionLayout(CommandBuffer& cmdBuffer, ImageLayout oldLayout, ImageLayout newLayout) const;
  ImageLayout layout() const { return mLayout; }
  void setLayoutAfterCopy(ImageLayout layout) const { mLayout = layout; }
  Format format() const { return mFormat; }
  Extent extent() const { return mExtent; }
  uint32_t pitch() const { return mPitch; }
  const Buffer& storage() const { return mStorage; }
  // tightly packed host rows <-> pitched device rows
  void upload(const void* tight) {
    const uint32_t row = mExtent.width * texelBytes(mFormat);
    std::vector<uint8_t> staged(size_t(mPitch) * mExtent.height, 0);
    for (uint32_t y = 0; y < mExtent.height; ++y) std::memcpy(staged.data() + size_t(y) * mPitch, static_cast<const uint8_t*>(tight) + size_t(y) * row, row);
    mStorage.upload(staged.data(), staged.size());
    check(vkr_queue_wait_idle_of(mStorage), "Image::upload");
  }
  void download(void* tight) const {
    const uint32_t row = mExtent.width * texelBytes(mFormat);
    std::vector<uint8_t> staged(size_t(mPitch) * mExtent.height);
    mStorage.download(staged.data(), staged.size());
    for (uint32_t y = 0; y < mExtent.height; ++y) std::memcpy(static_cast<uint8_t*>(tight) + size_t(y) * row, staged.data() + size_t(y) * mPitch, row);
  }

 private:
  static int32_t vkr_queue_wait_idle_of(const Buffer& b) {
    // a zero-byte download synchronises the owning stream without needing the context here
    uint8_t dummy;
    return vkr_buffer_download(b.getHandle(), 0, &dummy, 1);
  }
  Format mFormat;
  Extent mExtent;
  uint32_t mPitch;
  Buffer mStorage;
  mutable ImageLayout mLayout = ImageLayout::eUndefined;
};

class Sampler {};  // the deferred pass fetches texel centres with a nearest filter: nothing to configure

struct DescriptorImageInfo {
  const Sampler* sampler = nullptr;
  const ImageView* imageView = nullptr;
  ImageLayout imageLayout = ImageLayout::eUndefined;
  DescriptorImageInfo() = default;
  DescriptorImageInfo(const Sampler& s, const ImageView& v, ImageLayout l) : sampler{&s}, imageView{&v}, imageLayout{l} {}
  DescriptorImageInfo(std::nullptr_t, const ImageView& v, ImageLayout l) : imageView{&v}, imageLayout{l} {}
};

// ---------------------------------------------------------------- shaders / layouts / descriptors
struct ShaderIOBinding {
  ShaderIOBinding(DescriptorType type, uint32_t count = 1, uint32_t setIndex = 0, bool isVariable = false)
      : type{type}, count{count}, setIndex{setIndex}, isVariable{isVariable} {}
  DescriptorType type;
  uint32_t count, setIndex;
  bool isVariable;
};

// The SPIR-V words of vw::Shader become the name of a kernel compiled into the library.
class Shader {
 public:
  Shader(Device& device, ShaderStageFlagBits stage, const std::string& kernelName, ArrayProxy<ShaderIOBinding> ioBindings, uint32_t pushConstantSize = 0)
      : mIOBindings{ioBindings.begin(), ioBindings.end()}, mPushConstantSize{pushConstantSize}, mStage{stage}, mName{kernelName} {
    check(vkr_kernel_get(device.getHandle(), kernelName.c_str(), &mKernel), "vkr::Shader");
    uint32_t lo = 0, hi = 0;
    check(vkr_kernel_push_constant_size(mKernel, &lo, &hi), "vkr::Shader");
    if (pushConstantSize < lo || pushConstantSize > hi)
      throw std::runtime_error("vkr::Shader: kernel '" + kernelName + "' takes " + std::to_string(lo) + ".." + std::to_string(hi) + " bytes of push constants");
  }
  const std::vector<ShaderIOBinding>& getIOBindings() const { return mIOBindings; }
  uint32_t getPushConstantSize() const { return mPushConstantSize; }
  ShaderStageFlagBits getStage() const { return mStage; }
  vkr_kernel* getHandle() const { return mKernel; }

 private:
  std::vector<ShaderIOBinding> mIOBindings;
  uint32_t mPushConstantSize;
  ShaderStageFlagBits mStage;
  std::string mName;
  vkr_kernel* mKernel = nullptr;
};

struct DescriptorSetLayoutBinding {
  uint32_t binding;
  DescriptorType type;
  uint32_t count;
};

class DescriptorSet;
struct WriteDescriptorSet {
  DescriptorSet* dstSet;
  uint32_t dstBinding, dstArrayElement;
  DescriptorType type;
  std::vector<DescriptorImageInfo> images;
  std::vector<DescriptorBufferInfo> buffers;
};

class DescriptorSet {
 public:
  struct Slot {
    DescriptorType type;
    std::vector<DescriptorImageInfo> images;
    std::vector<DescriptorBufferInfo> buffers;
  };
  explicit DescriptorSet(const std::vector<DescriptorSetLayoutBinding>& bindings) {
    for (const auto& b : bindings) {
      Slot s{b.type, {}, {}};
      if (b.type == DescriptorType::eCombinedImageSampler || b.type == DescriptorType::eStorageImage)
        s.images.resize(b.count);
      else
        s.buffers.resize(b.count);
      mSlots.push_back(std::move(s));
    }
  }
  WriteDescriptorSet writeImages(uint32_t dstBinding, DescriptorType type, ArrayProxy<DescriptorImageInfo> infos, uint32_t arrayOffset = 0) {
    return {this, dstBinding, arrayOffset, type, {infos.begin(), infos.end()}, {}};
  }
  WriteDescriptorSet writeBuffers(uint32_t dstBinding, DescriptorType type, ArrayProxy<DescriptorBufferInfo> infos, uint32_t arrayOffset = 0) {
    return {this, dstBinding, arrayOffset, type, {}, {infos.begin(), infos.end()}};
  }
  void apply(const WriteDescriptorSet& w) {
    if (w.dstBinding >= mSlots.size()) throw std::runtime_error("updateDescriptorSets: binding out of range");
    Slot& s = mSlots[w.dstBinding];
    if (s.type != w.type) throw std::runtime_error("updateDescriptorSets: descriptor type does not match the layout");
    for (size_t i = 0; i < w.images.size(); ++i) {
      if (w.dstArrayElement + i >= s.images.size()) throw std::runtime_error("updateDescriptorSets: array element out of range");
      s.images[w.dstArrayElement + i] = w.images[i];
    }
    for (size_t i = 0; i < w.buffers.size(); ++i) {
      if (w.dstArrayElement + i >= s.buffers.size()) throw std::runtime_error("updateDescriptorSets: array element out of range");
      s.buffers[w.dstArrayElement + i] = w.buffers[i];
    }
  }
  const std::vector<Slot>& slots() const { return mSlots; }

 private:
  std::vector<Slot> mSlots;
};

// vk::Device::updateDescriptorSets(writes, copies)
inline void updateDescriptorSets(ArrayProxy<WriteDescriptorSet> writes) {
  for (const auto& w : writes) w.dstSet->apply(w);
}

class DedicatedDescriptorPool {
 public:
  DedicatedDescriptorPool(uint32_t setCount, const std::vector<DescriptorSetLayoutBinding>& bindings) {
    mSets.reserve(setCount);
    for (uint32_t i = 0; i < setCount; ++i) mSets.emplace_back(bindings);
  }
  std::vector<DescriptorSet>& getSets() { return mSets; }

 private:
  std::vector<DescriptorSet> mSets;
};

class DescriptorSetLayout {
 public:
  explicit DescriptorSetLayout(std::vector<DescriptorSetLayoutBinding> bindings) : mBindings{std::move(bindings)} {}
  DedicatedDescriptorPool createDedicatedPool(uint32_t setCount, uint32_t /*variableDescriptorCount*/ = 0) const { return {setCount, mBindings}; }
  const std::vector<DescriptorSetLayoutBinding>& bindings() const { return mBindings; }

 private:
  std::vector<DescriptorSetLayoutBinding> mBindings;
};

// Bindings are numbered per set in declaration order and push-constant ranges stack in shader
// order, exactly as vw::PipelineLayout does (vkdescriptor.cpp:43-79).
class PipelineLayout {
 public:
  PipelineLayout(ArrayProxy<std::reference_wrapper<Shader>> shaders) {
    std::vector<std::vector<DescriptorSetLayoutBinding>> perSet(1);
    for (const Shader& shader : shaders) {
      for (const ShaderIOBinding& io : shader.getIOBindings()) {
        if (io.setIndex >= perSet.size()) perSet.resize(io.setIndex + 1);
        auto& set = perSet[io.setIndex];
        set.push_back({static_cast<uint32_t>(set.size()), io.type, io.count});  // next free binding number of that set
      }
      mPushSize += shader.getPushConstantSize();
    }
    for (auto& s : perSet) mLayouts.emplace_back(std::move(s));
  }
  const std::vector<DescriptorSetLayout>& getDescLayouts() const { return mLayouts; }
  uint32_t pushConstantSize() const { return mPushSize; }

 private:
  std::vector<DescriptorSetLayout> mLayouts;
  uint32_t mPushSize = 0;
};

class ComputePipeline {
 public:
  ComputePipeline(const PipelineLayout& layout, const Shader& computeShader) : mLayout{&layout}, mKernel{computeShader.getHandle()} {
    if (computeShader.getStage() != ShaderStageFlagBits::eCompute) throw std::runtime_error("vkr::ComputePipeline: not a compute shader");
  }
  vkr_kernel* kernel() const { return mKernel; }
  const PipelineLayout* layout() const { return mLayout; }

 private:
  const PipelineLayout* mLayout;
  vkr_kernel* mKernel;
};

// ---------------------------------------------------------------- command recording / submission
class CommandBuffer {
 public:
  explicit CommandBuffer(vkr_context* ctx) : mCtx{ctx} {}
  CommandBuffer(const CommandBuffer&) = delete;
  CommandBuffer& operator=(const CommandBuffer&) = delete;
  CommandBuffer(CommandBuffer&& o) noexcept : mCtx{o.mCtx}, mCmd{o.mCmd}, mRecording{o.mRecording}, mInFlight{std::move(o.mInFlight)} { o.mCmd = nullptr; }
  ~CommandBuffer() { vkr_cmd_destroy(mCmd); }

  template <typename T>
  void record(T recordFunc) {
    vkr_cmd_destroy(mCmd);
    mCmd = nullptr;
    check(vkr_cmd_begin(mCtx, &mCmd), "CommandBuffer::begin");
    mRecording = true;
    try {
      recordFunc(*this);
    } catch (...) {
      mRecording = false;
      throw;
    }
    mRecording = false;
    check(vkr_cmd_end(mCmd), "CommandBuffer::end");
  }
  bool recording() const { return mRecording; }
  void bindPipeline(PipelineBindPoint, const ComputePipeline& pipeline) { check(vkr_cmd_bind_kernel(mCmd, pipeline.kernel()), "bindPipeline"); }
  void pushConstants(const PipelineLayout&, ShaderStageFlagBits, uint32_t offset, uint32_t size, const void* values) {
    check(vkr_cmd_push_constants(mCmd, offset, size, values), "pushConstants");
  }
  void bindDescriptorSets(PipelineBindPoint, const PipelineLayout&, uint32_t firstSet, std::initializer_list<std::reference_wrapper<DescriptorSet>> sets,
                          std::initializer_list<uint32_t> /*dynamicOffsets*/ = {}) {
    uint32_t set = firstSet;
    for (DescriptorSet& ds : sets) {
      uint32_t binding = 0;
      for (const auto& slot : ds.slots()) {
        for (uint32_t e = 0; e < slot.images.size(); ++e) {
          const ImageView* v = slot.images[e].imageView;
          if (v == nullptr) continue;
          const Image* im = v->image();
          check(vkr_cmd_bind_image2d(mCmd, set, binding, e, im->storage().getHandle(), 0, im->extent().width, im->extent().height, im->pitch(),
                                     static_cast<uint32_t>(im->format())),
                "bindDescriptorSets(image)");
        }
        for (uint32_t e = 0; e < slot.buffers.size(); ++e) {
          const DescriptorBufferInfo& b = slot.buffers[e];
          if (b.buffer == nullptr) continue;
          if (e != 0) throw std::runtime_error("bindDescriptorSets: buffer arrays are not supported");
          check(vkr_cmd_bind_buffer(mCmd, set, binding, b.buffer->getHandle(), b.offset, b.range), "bindDescriptorSets(buffer)");
        }
        ++binding;
      }
      ++set;
    }
  }
  void dispatch(uint32_t x, uint32_t y, uint32_t z) { check(vkr_cmd_dispatch(mCmd, x, y, z), "dispatch"); }

  // The buffer may be recorded again once the fence of its last submission has signalled
  // (the bookkeeping of vkcore.hpp:69-82).
  void submittedWith(std::shared_ptr<Fence> fence) { mInFlight = std::move(fence); }
  bool inFlight() {
    if (mInFlight && mInFlight->signaled()) mInFlight.reset();
    return mInFlight != nullptr;
  }
  vkr_cmd* getHandle() const { return mCmd; }

 private:
  vkr_context* mCtx;
  vkr_cmd* mCmd = nullptr;
  bool mRecording = false;
  std::shared_ptr<Fence> mInFlight;
};

inline void Image::transitionLayout(CommandBuffer& cmdBuffer, ImageLayout oldLayout, ImageLayout newLayout) const {
  if (!cmdBuffer.recording()) throw std::runtime_error("Image::transitionLayout: the command buffer is not recording");
  if (oldLayout != ImageLayout::eUndefined && oldLayout != mLayout)
    throw std::runtime_error("Image::transitionLayout: oldLayout is not the layout the image is in");
  mLayout = newLayout;
}

class Queue {
 public:
  explicit Queue(vkr_context* ctx) : mCtx{ctx} {}
  void allocateOneTimeBuffers(uint32_t count) {
    for (uint32_t i = 0; i < count; ++i) mBuffers.emplace_back(mCtx);
  }
  bool hasReadyBuffer() { return idleBuffer() != mBuffers.end(); }
  template <typename T>
  std::shared_ptr<Fence> oneTimeRecordSubmit(T recordFunc, std::initializer_list<std::reference_wrapper<Semaphore>> = {}, std::initializer_list<uint32_t> = {},
                                             std::initializer_list<std::reference_wrapper<Semaphore>> = {}) {
    const auto slot = idleBuffer();
    if (slot == mBuffers.end())
      throw std::runtime_error("Queue::oneTimeRecordSubmit: every one-time command buffer is still in flight (ask hasReadyBuffer() first)");
    slot->record(recordFunc);
    vkr_fence* f = nullptr;
    check(vkr_queue_submit(mCtx, slot->getHandle(), &f), "Queue::submit");
    auto fence = std::make_shared<Fence>(f);
    slot->submittedWith(fence);
    return fence;
  }
  void waitIdle() const { check(vkr_queue_wait_idle(mCtx), "Queue::waitIdle"); }

 private:
  std::vector<CommandBuffer>::iterator idleBuffer() {
    return std::find_if(mBuffers.begin(), mBuffers.end(), [](CommandBuffer& b) { return !b.inFlight(); });
  }
  vkr_context* mCtx;
  std::vector<CommandBuffer> mBuffers;
};

inline Queue& Device::getPreferredQueue() {
  if (!mQueue) mQueue = std::make_unique<Queue>(mCtx);
  return *mQueue;
}
inline Device::~Device() {
  mQueue.reset();
  vkr_context_destroy(mCtx);
}

// Pinned staging area filled front to back, with the copies out of it deferred to flush()
// (vw::StagingBuffer, vkmemory.hpp:116-219).
class StagingBuffer : public Buffer {
 public:
  StagingBuffer(MemoryAllocator& allocator, uint64_t size, Queue& transferQueue)
      : Buffer{allocator, size, BufferUse::kStagingBuffer, VKR_MEMORY_USAGE_CPU_TO_GPU}, mQueue{transferQueue} {}
  template <typename T>
  void queueBufferCopy(const T& src, const Buffer& dst, uint64_t dstOffset = 0) {
    const uint64_t bytes = byteSize(src);
    const uint64_t at = reserve(bytes, 1, "queueBufferCopy");
    std::memcpy(mMappedPtr + at, src.data(), bytes);
    mCopies.push_back({&dst, at, dstOffset, bytes});
  }
  // Several containers to consecutive destination offsets: one copy when they fit the staging area together,
  // else one copy each.
  template <typename T>
  void queueBufferCopies(const T& srcs, const Buffer& dst, uint64_t baseDstOffset = 0) {
    uint64_t total = 0;
    for (const auto& src : srcs) total += byteSize(src);
    if (total <= capacity()) {
      uint64_t at = reserve(total, 1, "queueBufferCopies");
      mCopies.push_back({&dst, at, baseDstOffset, total});
      for (const auto& src : srcs) {
        std::memcpy(mMappedPtr + at, src.data(), byteSize(src));
        at += byteSize(src);
      }
    } else {
      for (const auto& src : srcs) {
        queueBufferCopy(src, dst, baseDstOffset);
        baseDstOffset += byteSize(src);
      }
    }
  }
  // The file's tightly packed rows go into the staging area in the destination's row pitch, so that flush()
  // moves the image with one copy.  The layouts are recorded on the image at flush(), as the reference's two
  // transitions around its copy leave it in postLayout.
  void queueImageCopy(const ImageFile& imageFile, const Image& dst, ImageLayout /*preLayout*/ = ImageLayout::eUndefined,
                      ImageLayout postLayout = ImageLayout::eShaderReadOnlyOptimal) {
    const Extent e = imageFile.getExtent();
    if (e.width != dst.extent().width || e.height != dst.extent().height) throw std::runtime_error("StagingBuffer::queueImageCopy: extent differs from the image's");
    const uint64_t row = uint64_t(e.width) * texelBytes(dst.format());
    if (imageFile.dataSize() != row * e.height) throw std::runtime_error("StagingBuffer::queueImageCopy: data size does not match extent and format");
    const uint64_t bytes = uint64_t(dst.pitch()) * e.height;
    const uint64_t at = reserve(bytes, 16, "queueImageCopy");
    if (row == dst.pitch()) {
      imageFile.loadData(mMappedPtr + at);
    } else {
      std::vector<std::byte> tight(imageFile.dataSize());
      imageFile.loadData(tight.data());
      std::memset(mMappedPtr + at, 0, bytes);
      for (uint32_t y = 0; y < e.height; ++y) std::memcpy(mMappedPtr + at + uint64_t(y) * dst.pitch(), tight.data() + uint64_t(y) * row, row);
    }
    mCopies.push_back({&dst.storage(), at, 0, bytes});
    mImageLayouts.push_back({&dst, postLayout});
  }
  uint64_t remainingSpace() const { return capacity() - mUsedBytes; }
  void flush() {
    for (const auto& c : mCopies) check(vkr_buffer_upload(c.dst->getHandle(), c.dstOffset, mMappedPtr + c.srcOffset, c.bytes), "StagingBuffer::flush");
    mQueue.waitIdle();
    for (const auto& [image, layout] : mImageLayouts) image->setLayoutAfterCopy(layout);
    mCopies.clear();
    mImageLayouts.clear();
    mUsedBytes = 0;
  }

 private:
  uint64_t capacity() const { return mSegmentSizes[0]; }
  // Offset of `bytes` free bytes at the given alignment, flushing first when the tail is too short.
  uint64_t reserve(uint64_t bytes, uint64_t alignment, const char* who) {
    if (bytes > capacity()) throw std::runtime_error(std::string("StagingBuffer::") + who + ": " + std::to_string(bytes) + " bytes do not fit a staging area of " + std::to_string(capacity()));
    uint64_t at = mUsedBytes;
    alignTo(at, alignment);
    if (at + bytes > capacity()) {
      flush();
      at = 0;
    }
    mUsedBytes = at + bytes;
    return at;
  }
  struct Copy {
    const Buffer* dst;
    uint64_t srcOffset, dstOffset, bytes;
  };
  std::vector<Copy> mCopies;
  std::vector<std::pair<const Image*, ImageLayout>> mImageLayouts;
  uint64_t mUsedBytes = 0;
  Queue& mQueue;
};

// ---------------------------------------------------------------- callers of the two passes
using Mat4 = std::array<float, 16>;  // column-major, like glm::mat4

// glm::perspective as /root/reference/src/main.cpp:95 calls it.
inline Mat4 perspective(float fovyRadians, float aspect, float zNear, float zFar) {
  Mat4 m;
  vkr_perspective(fovyRadians, aspect, zNear, zFar, m.data());
  return m;
}

// Key and action codes with GLFW's values, so that a window callback can forward them unchanged.
namespace Key {
constexpr int eSpace = 32, eA = 65, eD = 68, eS = 83, eW = 87, eLeftShift = 340;
constexpr int eRelease = 0, ePress = 1, eRepeat = 2;
}  // namespace Key

// CameraInputHandler of /root/reference/inc/vkcamera.hpp:9-67 (same member functions; GLFW types dropped).
class CameraInputHandler {
 public:
  CameraInputHandler() { vkr_camera_default(&mCamera); }
  void onCursorPosChange(double x, double y) {
    const float nx = static_cast<float>(x), ny = static_cast<float>(y);
    if (mHasPrevCursorPos) vkr_camera_cursor_delta(&mCamera, nx - mPrevCursorPos[0], ny - mPrevCursorPos[1]);
    mPrevCursorPos = {nx, ny};
    mHasPrevCursorPos = true;
  }
  void onKeyEvent(int key, int action, [[maybe_unused]] int mods = 0) {
    if (action == Key::eRelease) return;
    int32_t k = -1;
    switch (key) {
      case Key::eW: k = VKR_KEY_W; break;
      case Key::eA: k = VKR_KEY_A; break;
      case Key::eS: k = VKR_KEY_S; break;
      case Key::eD: k = VKR_KEY_D; break;
      case Key::eSpace: k = VKR_KEY_SPACE; break;
      case Key::eLeftShift: k = VKR_KEY_LEFT_SHIFT; break;
      default: return;
    }
    vkr_camera_key(&mCamera, k, static_cast<float>(mDeltaTime));
  }
  void setLastDeltaTime(int64_t microseconds) { mDeltaTime = static_cast<double>(microseconds) / 10000.0; }
  Mat4 getView() const {
    Mat4 m;
    vkr_camera_view(&mCamera, m.data());
    return m;
  }
  std::array<float, 3> getPos() const { return {mCamera.pos[0], mCamera.pos[1], mCamera.pos[2]}; }
  std::array<float, 3> getForward() const { return {mCamera.forward[0], mCamera.forward[1], mCamera.forward[2]}; }
  const vkr_camera& state() const { return mCamera; }

 private:
  vkr_camera mCamera;
  double mDeltaTime = 1.0;
  std::array<float, 2> mPrevCursorPos{};
  bool mHasPrevCursorPos = false;
};

// view, vp = proj * view, inverse(vp), DeferredPushData and the cull planes for one frame
// (/root/reference/src/main.cpp:206-209).
inline vkr_frame_constants frameConstants(const CameraInputHandler& camera, const Mat4& proj) {
  vkr_frame_constants fc;
  check(vkr_frame_setup(&camera.state(), proj.data(), &fc), "vkr::frameConstants");
  return fc;
}

// The point lights of a .fscene file as LightInfo records (/root/reference/src/main.cpp:40-70).
inline std::vector<vkr_light_info> loadPointLights(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("File " + path + " does not exist!");
  const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  uint32_t n = 0;
  check(vkr_fscene_point_lights(text.data(), text.size(), nullptr, 0, &n), "vkr::loadPointLights");
  std::vector<vkr_light_info> lights(n);
  if (n) check(vkr_fscene_point_lights(text.data(), text.size(), lights.data(), n, &n), "vkr::loadPointLights");
  return lights;
}

inline vkr_image2d imageDesc(const Image& im) {
  return {im.storage().devicePtr(), im.extent().width, im.extent().height, im.pitch(), static_cast<uint32_t>(im.format())};
}

// Stands in for the raster G-buffer pass of /root/reference/src/main.cpp:214-219 in tests and benches: an
// analytic scene written into the four attachments (stream-ordered with the queue's submissions).
inline void fillGBuffer(Device& device, const Image& albedo, const Image& material, const Image& normal, const Image& depth, const Buffer& spheres,
                        const vkr_gbuffer_scene& scene) {
  vkr_gbuffer_fill_args a{};
  a.albedo = imageDesc(albedo);
  a.material = imageDesc(material);
  a.normal = imageDesc(normal);
  a.depth = imageDesc(depth);
  a.spheres = spheres.devicePtr();
  a.scene = scene;
  check(vkr_gbuffer_fill(device.getHandle(), &a), "vkr::fillGBuffer");
}

// The tables vw::Scene::Scene builds (/root/reference/src/vkmodel.cpp:55-111) from a flattened scene description.
struct SceneTables {
  vkr_scene_counts counts{};
  std::vector<vkr_mesh_info> meshInfo;
  std::vector<vkr_draw_indexed_indirect_command> drawCommands;
  std::vector<vkr_per_mesh_data> perMeshData;
  std::vector<float> meshMatrices;
  std::vector<uint32_t> meshBase, vertexBase, sourceMesh;

  explicit SceneTables(const vkr_scene_desc& desc) {
    check(vkr_scene_count(&desc, &counts), "vkr::SceneTables");
    meshInfo.resize(counts.kept_meshes);
    drawCommands.resize(counts.kept_meshes);
    perMeshData.resize(counts.kept_meshes);
    meshMatrices.resize(size_t(counts.kept_instances) * 16);
    meshBase.resize(counts.kept_meshes + 1);
    vertexBase.resize(counts.kept_meshes + 1);
    sourceMesh.resize(counts.kept_meshes);
    vkr_scene_tables t{meshInfo.data(), drawCommands.data(), perMeshData.data(), meshMatrices.data(), meshBase.data(), vertexBase.data(), sourceMesh.data()};
    check(vkr_scene_build_tables(&desc, &counts, &t), "vkr::SceneTables");
  }
};

}  // namespace vkr
