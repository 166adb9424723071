// Host program written the way /root/reference/src/main.cpp:88-92,109-117,139,156-173,187-231 is,
// against the vkr:: facade: build the deferred compute pipeline, bind the G-buffer / light UBO /
// output image through descriptor sets, push DeferredPushData, dispatch(W, H, 1) from a one-time
// command buffer, wait on the fence.  Then the same for the cull pass.  Inputs and outputs are raw
// files in the directory given on the command line (written and checked by tests/test_facade_gpu.py).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "vkr/vkr.hpp"

template <typename T>
static std::vector<T> loadBinaryFile(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("File " + path + " does not exist!");
  const std::streamsize n = f.tellg();
  if (n % sizeof(T) != 0) throw std::runtime_error("File " + path + " size is not a multiple of sizeof(T)");
  std::vector<T> v(n / sizeof(T));
  f.seekg(0);
  f.read(reinterpret_cast<char*>(v.data()), n);
  return v;
}
template <typename T>
static void saveBinaryFile(const std::string& path, const std::vector<T>& v) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), v.size() * sizeof(T));
}

struct alignas(float) LightInfo {  // src/main.cpp:40-47
  float pos[3];
  float opening_angle;
  float color[3];
  float penumbra_angle;
  float dir[3];
  float _temp;
};
struct DeferredPushData {  // src/main.cpp:76-80
  float cameraPos[3];
  float pointLightRadius;
  float inverseVP[16];
};
static_assert(sizeof(LightInfo) == 48 && sizeof(DeferredPushData) == 80);

// vw::ImageFile over a file of tightly packed texels (stands in for the reference's stb / DDS loaders)
class RawImageFile : public vkr::ImageFile {
 public:
  RawImageFile(const std::string& path, vkr::Extent extent) : mBytes{loadBinaryFile<uint8_t>(path)}, mExtent{extent} {}
  size_t dataSize() const override { return mBytes.size(); }
  void loadData(std::byte* dst) const override { std::memcpy(dst, mBytes.data(), mBytes.size()); }
  vkr::Extent getExtent() const override { return mExtent; }

 private:
  std::vector<uint8_t> mBytes;
  vkr::Extent mExtent;
};

struct CullPushData {
  float planes[24];
  uint32_t meshCount, firstMesh, firstInstance, flags;
};

int main(int argc, char** argv) {
  try {
    if (argc < 2) throw std::runtime_error("usage: facade_frame <dir>");
    const std::string dir = std::string(argv[1]) + "/";
    const auto meta = loadBinaryFile<uint32_t>(dir + "meta.bin");  // W, H, frames, N, M, cullFlags
    vkr::Extent windowExtent{meta[0], meta[1]};
    const uint32_t frameCount = meta[2];

    vkr::Device device{0};
    auto& queue = device.getPreferredQueue();
    vkr::MemoryAllocator allocator{device};

    std::vector<LightInfo> lightInfos = loadBinaryFile<LightInfo>(dir + "lights.bin");
    const auto pushWords = loadBinaryFile<float>(dir + "push.bin");
    DeferredPushData deferredPush;
    std::memcpy(&deferredPush, pushWords.data(), sizeof(deferredPush));

    // src/main.cpp:90: the path the reference loads its SPIR-V from names the kernel here
    vkr::Shader deferredCompShader{device,
                                   vkr::ShaderStageFlagBits::eCompute,
                                   vkr::loadShader("shaders/deferred.comp.spv"),
                                   {{vkr::DescriptorType::eCombinedImageSampler, 4}, {vkr::DescriptorType::eUniformBuffer}, {vkr::DescriptorType::eStorageImage, 1, 1}},
                                   sizeof(DeferredPushData)};

    vkr::Buffer ubo{allocator, vkr::byteSize(lightInfos), vkr::BufferUse::kUniformBuffer, vkr::VKR_MEMORY_USAGE_CPU_TO_GPU};

    vkr::Image gAlbedo{allocator, vkr::Format::eR8G8B8A8Unorm, windowExtent};
    vkr::Image gSpecular{allocator, vkr::Format::eR8G8B8A8Unorm, windowExtent};
    vkr::Image gNormal{allocator, vkr::Format::eR16G16B16A16Sfloat, windowExtent};
    vkr::Image depthAttachment{allocator, vkr::Format::eD32Sfloat, windowExtent};
    // the G-buffer fill is a raster pass in the reference; here the planes come from files, staged the way the
    // reference stages its textures (vw::StagingBuffer::queueImageCopy + flush, src/vkmodel.cpp / src/main.cpp:102-107);
    // the staging area is smaller than the four planes together, so it flushes by itself on the way
    {
      const uint64_t px = uint64_t(windowExtent.width) * windowExtent.height;
      vkr::StagingBuffer gbufferStaging{allocator, px * 8 + px * 4 + 4096, queue};
      RawImageFile albedoFile{dir + "albedo.bin", windowExtent}, materialFile{dir + "material.bin", windowExtent};
      RawImageFile normalFile{dir + "normal.bin", windowExtent}, depthFile{dir + "depth.bin", windowExtent};
      gbufferStaging.queueImageCopy(albedoFile, gAlbedo);
      gbufferStaging.queueImageCopy(materialFile, gSpecular);
      gbufferStaging.queueImageCopy(normalFile, gNormal);
      gbufferStaging.queueImageCopy(depthFile, depthAttachment);
      gbufferStaging.flush();
      if (gNormal.layout() != vkr::ImageLayout::eShaderReadOnlyOptimal) throw std::runtime_error("queueImageCopy did not leave the image in its post layout");
    }

    auto gAlbedoView = gAlbedo.createView();
    auto gNormalView = gNormal.createView();
    auto gSpecularView = gSpecular.createView();
    auto depthAttachmentView = depthAttachment.createView();
    vkr::Sampler nearSampler;

    vkr::PipelineLayout deferredCompPipelineLayout{{deferredCompShader}};
    vkr::ComputePipeline deferredComputePipeline{deferredCompPipelineLayout, deferredCompShader};

    auto deferredDescriptorPool = deferredCompPipelineLayout.getDescLayouts()[0].createDedicatedPool(1);
    auto& deferredDescriptorSet = deferredDescriptorPool.getSets()[0];

    vkr::DescriptorImageInfo deferredDescriptorImageInfos[] = {{nearSampler, gAlbedoView, vkr::ImageLayout::eShaderReadOnlyOptimal},
                                                               {nearSampler, gSpecularView, vkr::ImageLayout::eShaderReadOnlyOptimal},
                                                               {nearSampler, gNormalView, vkr::ImageLayout::eShaderReadOnlyOptimal},
                                                               {nearSampler, depthAttachmentView, vkr::ImageLayout::eShaderReadOnlyOptimal}};
    vkr::DescriptorBufferInfo deferredDescriptorUboInfo{&ubo, 0, vkr::byteSize(lightInfos)};
    vkr::updateDescriptorSets({deferredDescriptorSet.writeImages(0, vkr::DescriptorType::eCombinedImageSampler, deferredDescriptorImageInfos),
                               deferredDescriptorSet.writeBuffers(1, vkr::DescriptorType::eUniformBuffer, deferredDescriptorUboInfo)});

    // three "swapchain" images the compute pass writes in turn
    const uint32_t swapImageCount = 3;
    std::vector<vkr::Image> swapImages;
    swapImages.reserve(swapImageCount);
    for (uint32_t i = 0; i < swapImageCount; ++i) swapImages.emplace_back(allocator, vkr::Format::eR8G8B8A8Unorm, windowExtent);
    std::vector<vkr::ImageView> swapViews;
    for (auto& im : swapImages) swapViews.push_back(im.createView());
    auto swapImageDescriptorPool = deferredCompPipelineLayout.getDescLayouts()[1].createDedicatedPool(swapImageCount);
    auto& swapImageDescriptorSets = swapImageDescriptorPool.getSets();
    std::vector<vkr::WriteDescriptorSet> swapDescriptorWrites;
    for (uint32_t i = 0; i < swapImageCount; ++i)
      swapDescriptorWrites.push_back(swapImageDescriptorSets[i].writeImages(0, vkr::DescriptorType::eStorageImage, vkr::DescriptorImageInfo{nullptr, swapViews[i], vkr::ImageLayout::eGeneral}));
    vkr::updateDescriptorSets(swapDescriptorWrites);

    queue.allocateOneTimeBuffers(swapImageCount);
    uint32_t submitted = 0, skipped = 0;
    for (uint32_t frame = 0; submitted < frameCount; ++frame) {
      ubo.copyToMapped(lightInfos);
      if (!queue.hasReadyBuffer()) {
        ++skipped;
        continue;
      }
      const uint32_t imageIndex = submitted % swapImageCount;
      // src/main.cpp:221-227, the compute half of the record lambda, with vw:: -> vkr:: and swapchain.getImage(i) ->
      // swapImages[i]; lines :214-219 before it are the raster pass
      queue.oneTimeRecordSubmit([&](vkr::CommandBuffer& commandBuffer) {
        vkr::Image::transitionLayout(commandBuffer, swapImages[imageIndex], vkr::ImageLayout::eUndefined, vkr::ImageLayout::eGeneral);
        commandBuffer.bindPipeline(vkr::PipelineBindPoint::eCompute, deferredComputePipeline);
        commandBuffer.pushConstants(deferredCompPipelineLayout, vkr::ShaderStageFlagBits::eCompute, 0, sizeof(deferredPush), &deferredPush);
        commandBuffer.bindDescriptorSets(vkr::PipelineBindPoint::eCompute, deferredCompPipelineLayout, 0,
                                         {deferredDescriptorSet, swapImageDescriptorSets[imageIndex]}, {});
        commandBuffer.dispatch(windowExtent.width, windowExtent.height, 1);
        vkr::Image::transitionLayout(commandBuffer, swapImages[imageIndex], vkr::ImageLayout::eGeneral, vkr::ImageLayout::ePresentSrcKHR);
      });
      ++submitted;
    }
    device.waitIdle();
    for (uint32_t i = 0; i < swapImageCount && i < frameCount; ++i) {
      std::vector<uint8_t> out(size_t(windowExtent.width) * windowExtent.height * 4);
      swapImages[i].download(out.data());
      saveBinaryFile(dir + "out" + std::to_string(i) + ".bin", out);
    }

    // ---- cull pass: same surface, kernel "prep_indirect" -------------------------------------
    const uint32_t N = meta[3], M = meta[4];
    if (N > 0) {
      auto aabbMin = loadBinaryFile<float>(dir + "aabb_min.bin");
      auto aabbMax = loadBinaryFile<float>(dir + "aabb_max.bin");
      auto meshBase = loadBinaryFile<uint32_t>(dir + "mesh_base.bin");
      auto cmdTemplate = loadBinaryFile<vkr_draw_indexed_indirect_command>(dir + "cmd_template.bin");
      auto planes = loadBinaryFile<float>(dir + "planes.bin");

      vkr::Shader cullShader{device,
                             vkr::ShaderStageFlagBits::eCompute,
                             vkr::loadShader("shaders/prep_indirect.comp.spv"),
                             {{vkr::DescriptorType::eStorageBuffer}, {vkr::DescriptorType::eStorageBuffer}, {vkr::DescriptorType::eStorageBuffer},
                              {vkr::DescriptorType::eStorageBuffer}, {vkr::DescriptorType::eStorageBuffer}, {vkr::DescriptorType::eStorageBuffer}},
                             sizeof(CullPushData)};
      vkr::PipelineLayout cullLayout{{cullShader}};
      vkr::ComputePipeline cullPipeline{cullLayout, cullShader};
      // one multi-segment storage buffer, segments 256-byte aligned like vw::Buffer (vkmemory.cpp:23-30)
      vkr::Buffer ssbo{allocator,
                       {vkr::byteSize(aabbMin), vkr::byteSize(aabbMax), vkr::byteSize(meshBase), vkr::byteSize(cmdTemplate), vkr::byteSize(cmdTemplate), uint64_t(N) * 4},
                       vkr::BufferUse::kStorageBuffer};
      vkr::StagingBuffer staging{allocator, 64u << 20, queue};
      staging.queueBufferCopy(aabbMin, ssbo, ssbo.getSegmentDesc(0).offset);
      staging.queueBufferCopy(aabbMax, ssbo, ssbo.getSegmentDesc(1).offset);
      {  // the segment table in two pieces to consecutive offsets with one call (vw::StagingBuffer::queueBufferCopies)
        const size_t half = meshBase.size() / 2;
        std::vector<std::vector<uint32_t>> pieces{{meshBase.begin(), meshBase.begin() + half}, {meshBase.begin() + half, meshBase.end()}};
        staging.queueBufferCopies(pieces, ssbo, ssbo.getSegmentDesc(2).offset);
      }
      staging.queueBufferCopy(cmdTemplate, ssbo, ssbo.getSegmentDesc(3).offset);
      staging.flush();

      auto cullPool = cullLayout.getDescLayouts()[0].createDedicatedPool(1);
      auto& cullSet = cullPool.getSets()[0];
      std::vector<vkr::WriteDescriptorSet> writes;
      for (uint32_t b = 0; b < 6; ++b) writes.push_back(cullSet.writeBuffers(b, vkr::DescriptorType::eStorageBuffer, ssbo.getSegmentDesc(b)));
      vkr::updateDescriptorSets(writes);

      CullPushData cullPush{};
      std::memcpy(cullPush.planes, planes.data(), sizeof(cullPush.planes));
      cullPush.meshCount = M;
      cullPush.flags = meta[5];
      auto fence = queue.oneTimeRecordSubmit([&](vkr::CommandBuffer& cb) {
        cb.bindPipeline(vkr::PipelineBindPoint::eCompute, cullPipeline);
        cb.pushConstants(cullLayout, vkr::ShaderStageFlagBits::eCompute, 0, sizeof(cullPush), &cullPush);
        cb.bindDescriptorSets(vkr::PipelineBindPoint::eCompute, cullLayout, 0, {cullSet}, {});
        cb.dispatch(N, 1, 1);
      });
      fence->wait();
      if (!fence->signaled()) throw std::runtime_error("fence not signalled after wait");
      std::vector<vkr_draw_indexed_indirect_command> cmds(M);
      std::vector<uint32_t> visible(N);
      ssbo.download(cmds.data(), vkr::byteSize(cmds), ssbo.getSegmentDesc(4).offset);
      ssbo.download(visible.data(), vkr::byteSize(visible), ssbo.getSegmentDesc(5).offset);
      saveBinaryFile(dir + "cmds.bin", cmds);
      saveBinaryFile(dir + "visible.bin", visible);
    }

    // error paths of the layout bookkeeping: a transition from a layout the image is not in
    {
      bool layoutThrew = false;
      try {
        queue.oneTimeRecordSubmit([&](vkr::CommandBuffer& cb) {
          vkr::Image::transitionLayout(cb, swapImages[0], vkr::ImageLayout::eShaderReadOnlyOptimal, vkr::ImageLayout::eGeneral);
        });
      } catch (const std::runtime_error&) {
        layoutThrew = true;
      }
      if (!layoutThrew) throw std::runtime_error("a transition from the wrong layout was accepted");
    }

    // error path: unknown kernel -> std::runtime_error, like the reference's wrapper-level failures
    bool threw = false;
    try {
      vkr::Shader bogus{device, vkr::ShaderStageFlagBits::eCompute, "offscreen", {}, 0};
    } catch (std::runtime_error& err) {
      threw = std::string(err.what()).find("no kernel named") != std::string::npos;
    }
    if (!threw) throw std::runtime_error("unknown kernel name did not raise");

    std::cout << "facade ok: " << device.description() << " frames=" << submitted << " skipped=" << skipped << std::endl;
  } catch (std::runtime_error& err) {
    std::cout << "std::runtime_error: " << err.what() << std::endl;
    return 1;
  }
  return 0;
}
