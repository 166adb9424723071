// The frame loop of /root/reference/src/main.cpp:198-231 driven by a moving camera, written against the
// vkr:: facade: lights from a .fscene file (:48-70), CameraInputHandler (vkcamera.hpp) moved by key and
// cursor events, per-frame view / vp / inverse (:206-209), light UBO rewritten every frame (:202),
// hasReadyBuffer back-pressure with three command buffers (:200,203), the deferred pass dispatched per
// frame (:222-226).  The raster G-buffer pass is replaced by the analytic fill.  Inputs and outputs are
// files in the directory given on the command line (tests/test_facade_gpu.py writes and checks them).
#include <cmath>
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

struct DeferredPushData {  // src/main.cpp:76-80
  float cameraPos[3];
  float pointLightRadius;
  float inverseVP[16];
};

int main(int argc, char** argv) {
  try {
    if (argc < 2) throw std::runtime_error("usage: camera_path <dir>");
    const std::string dir = std::string(argv[1]) + "/";
    const auto meta = loadBinaryFile<uint32_t>(dir + "meta.bin");  // W, H, frames
    vkr::Extent windowExtent{meta[0], meta[1]};
    const uint32_t frameCount = meta[2];

    vkr::Device device{0};
    auto& queue = device.getPreferredQueue();
    vkr::MemoryAllocator allocator{device};

    std::vector<vkr_light_info> lightInfos = vkr::loadPointLights(dir + "scene.fscene");
    if (lightInfos.empty()) throw std::runtime_error("no point lights in scene.fscene");
    const auto spheresHost = loadBinaryFile<vkr_sphere>(dir + "spheres.bin");

    vkr::Shader deferredCompShader{device,
                                   vkr::ShaderStageFlagBits::eCompute,
                                   "deferred",
                                   {{vkr::DescriptorType::eCombinedImageSampler, 4}, {vkr::DescriptorType::eUniformBuffer}, {vkr::DescriptorType::eStorageImage, 1, 1}},
                                   sizeof(DeferredPushData)};
    vkr::Mat4 proj = vkr::perspective(70.0f * 0.01745329251994329576923690768489f, static_cast<float>(windowExtent.width) / static_cast<float>(windowExtent.height), 0.1f, 10000.0f);

    // device-resident light table, restaged each frame from one pinned copy per command buffer
    const uint32_t swapImageCount = 3;
    vkr::Buffer ubo{allocator, vkr::byteSize(lightInfos), vkr::BufferUse::kUniformBuffer};
    std::vector<vkr::Buffer> uboStaging;
    for (uint32_t i = 0; i < swapImageCount; ++i) uboStaging.emplace_back(allocator, vkr::byteSize(lightInfos), vkr::BufferUse::kStagingBuffer, vkr::VKR_MEMORY_USAGE_CPU_TO_GPU);
    vkr::Buffer spheres{allocator, vkr::byteSize(spheresHost), vkr::BufferUse::kStorageBuffer};
    spheres.upload(spheresHost.data(), vkr::byteSize(spheresHost));
    queue.waitIdle();

    vkr::Image gAlbedo{allocator, vkr::Format::eR8G8B8A8Unorm, windowExtent};
    vkr::Image gSpecular{allocator, vkr::Format::eR8G8B8A8Unorm, windowExtent};
    vkr::Image gNormal{allocator, vkr::Format::eR16G16B16A16Sfloat, windowExtent};
    vkr::Image depthAttachment{allocator, vkr::Format::eD32Sfloat, windowExtent};
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

    vkr::CameraInputHandler camera;
    camera.setLastDeltaTime(25000);  // 2.5 units of 10 ms: W moves 0.5 per frame
    double cursorX = 100.0;
    camera.onCursorPosChange(cursorX, 50.0);  // first event only records the position
    std::vector<float> cameraLog;

    queue.allocateOneTimeBuffers(swapImageCount);
    uint32_t submitted = 0, skipped = 0;
    while (submitted < frameCount) {
      if (!queue.hasReadyBuffer()) {
        ++skipped;
        continue;
      }
      const uint32_t imageIndex = submitted % swapImageCount;
      uboStaging[imageIndex].copyToMapped(lightInfos);
      void* mapped = nullptr;
      vkr::check(vkr_buffer_map(uboStaging[imageIndex].getHandle(), &mapped), "map");
      ubo.upload(mapped, vkr::byteSize(lightInfos));

      const vkr_frame_constants fc = vkr::frameConstants(camera, proj);
      DeferredPushData deferredPush;
      std::memcpy(&deferredPush, fc.deferred_push, sizeof(deferredPush));
      for (float v : camera.getPos()) cameraLog.push_back(v);
      for (float v : camera.getForward()) cameraLog.push_back(v);

      vkr_gbuffer_scene scene{};
      std::memcpy(scene.view_proj, fc.view_proj, 64);
      std::memcpy(scene.inv_view_proj, fc.inv_view_proj, 64);
      std::memcpy(scene.camera_pos, fc.deferred_push, 12);
      scene.sphere_count = vkr::size32(spheresHost);
      const float plane[4] = {0.0f, 1.0f, 0.0f, 0.0f}, colA[3] = {0.8f, 0.8f, 0.8f}, colB[3] = {0.2f, 0.2f, 0.25f};
      std::memcpy(scene.ground_plane, plane, 16);
      std::memcpy(scene.ground_albedo_a, colA, 12);
      std::memcpy(scene.ground_albedo_b, colB, 12);
      scene.ground_tile = 4.0f;
      scene.ground_roughness = 0.6f;
      scene.ground_metalness = 0.0f;
      vkr::fillGBuffer(device, gAlbedo, gSpecular, gNormal, depthAttachment, spheres, scene);

      queue.oneTimeRecordSubmit([&](vkr::CommandBuffer& commandBuffer) {
        commandBuffer.bindPipeline(vkr::PipelineBindPoint::eCompute, deferredComputePipeline);
        commandBuffer.pushConstants(deferredCompPipelineLayout, vkr::ShaderStageFlagBits::eCompute, 0, sizeof(deferredPush), &deferredPush);
        commandBuffer.bindDescriptorSets(vkr::PipelineBindPoint::eCompute, deferredCompPipelineLayout, 0, {deferredDescriptorSet, swapImageDescriptorSets[imageIndex]}, {});
        commandBuffer.dispatch(windowExtent.width, windowExtent.height, 1);
      });
      ++submitted;

      camera.onKeyEvent(vkr::Key::eW, vkr::Key::ePress);
      cursorX += 1.0 / 60.0;  // one degree of yaw per frame
      camera.onCursorPosChange(cursorX, 50.0);
      camera.onKeyEvent(vkr::Key::eW, vkr::Key::eRelease);  // ignored
    }
    device.waitIdle();
    for (uint32_t i = 0; i < swapImageCount && i < frameCount; ++i) {
      std::vector<uint8_t> out(size_t(windowExtent.width) * windowExtent.height * 4);
      swapImages[i].download(out.data());
      saveBinaryFile(dir + "out" + std::to_string(i) + ".bin", out);
    }
    saveBinaryFile(dir + "camera_log.bin", cameraLog);
    std::cout << "camera path ok: " << device.description() << " frames=" << submitted << " lights=" << lightInfos.size() << " skipped=" << skipped << std::endl;
  } catch (std::runtime_error& err) {
    std::cout << "std::runtime_error: " << err.what() << std::endl;
    return 1;
  }
  return 0;
}
