// rt_present.cxx — UNTESTED, never compiled (no Windows toolchain here; see README.md).
// Written against the reference at v1: gD3d.device / gD3d.device_context (gfx/gfx.cpp:4-11, used
// as in imgui/imgui.cpp:16), d3d_check_hresult_ / d3d_safe_release_ (gfx/d3d_layer.hxx:19-37),
// alloc_perm (base/memory.cxx:23-34), errf (first.cpp:27-43), ImGui 1.89.6 with the DX11 backend
// whose ImTextureID is an ID3D11ShaderResourceView* (imgui/imgui_dx11.hxx:5).
#include "rt_present.hxx"

namespace rt {
Rt_Present gRt_Present = {};

void rt_present_init_or_panic(u32 width, u32 height) {
  auto &P = gRt_Present;
  P.width  = width;
  P.height = height;
  P.rgba   = (u32 *)alloc_perm((s64)width * height * (s64)sizeof(u32));

  ::D3D11_TEXTURE2D_DESC desc = {};
  desc.Width            = width;
  desc.Height           = height;
  desc.MipLevels        = 1;
  desc.ArraySize        = 1;
  desc.Format           = DXGI_FORMAT_R8G8B8A8_UNORM;
  desc.SampleDesc.Count = 1;
  desc.Usage            = D3D11_USAGE_DYNAMIC;          // CPU writes it every frame: Map(WRITE_DISCARD)
  desc.BindFlags        = D3D11_BIND_SHADER_RESOURCE;
  desc.CPUAccessFlags   = D3D11_CPU_ACCESS_WRITE;
  ::HRESULT hr = gD3d.device->CreateTexture2D(&desc, NULL, &P.texture);
  d3d_check_hresult_(hr);

  ::D3D11_SHADER_RESOURCE_VIEW_DESC view_desc = {};
  view_desc.Format              = desc.Format;
  view_desc.ViewDimension       = D3D11_SRV_DIMENSION_TEXTURE2D;
  view_desc.Texture2D.MipLevels = 1;
  hr = gD3d.device->CreateShaderResourceView(P.texture, &view_desc, &P.view);
  d3d_check_hresult_(hr);

  P.cam_from   = {13.f, 2.f, 3.f};
  P.cam_at     = {0.f, 0.f, 0.f};
  P.vfov_deg   = 20.f;
  P.aperture   = 0.f;
  P.focus_dist = 10.f;
  P.settings   = rt_default_settings(width, height, 1, RT_MAX_DEPTH);   // progressive: 1 spp per displayed frame
  P.restart    = true;
}

void rt_present_shutdown() {
  d3d_safe_release_(gRt_Present.view);
  d3d_safe_release_(gRt_Present.texture);
}

void rt_present_upload(u32 const *rgba8) {
  auto &P = gRt_Present;
  ::D3D11_MAPPED_SUBRESOURCE mapped = {};
  ::HRESULT hr = gD3d.device_context->Map(P.texture, 0, D3D11_MAP_WRITE_DISCARD, 0, &mapped);
  d3d_check_hresult_(hr);
  for (u32 y = 0; y < P.height; y++) {
    u32 const *src = rgba8 + (s64)(P.height - 1 - y) * P.width;          // frame row 0 is the bottom row
    u8        *dst = (u8 *)mapped.pData + (s64)y * mapped.RowPitch;
    for (u32 x = 0; x < P.width; x++) {
      u32 const c = src[x];                                              // 0xRRGGBBAA (gfx.hpp:4-6)
      dst[4 * x + 0] = (u8)(c >> 24);
      dst[4 * x + 1] = (u8)(c >> 16);
      dst[4 * x + 2] = (u8)(c >> 8);
      dst[4 * x + 3] = (u8)c;
    }
  }
  gD3d.device_context->Unmap(P.texture, 0);
}

void rt_present_frame_or_panic() {
  auto &P = gRt_Present;
  if (P.restart) {
    P.camera = rt_camera_look_at(P.cam_from, P.cam_at, {0.f, 1.f, 0.f}, P.vfov_deg,
                                 (f32)P.width / (f32)P.height, P.aperture, P.focus_dist);
    P.settings.sample_offset = 0;
    P.settings.flags &= ~(u32)RTGPU_FLAG_ACCUMULATE;     // this frame starts the sum on the GPU
    P.samples_shown = 0;
    P.restart = false;
  }
  Render_Stats stats = {};
  rt_render_resolved_or_panic(P.camera, P.settings, P.rgba, &stats);
  P.settings.sample_offset += P.settings.spp;
  P.settings.flags |= RTGPU_FLAG_ACCUMULATE;
  P.samples_shown += P.settings.spp;
  f32 const ms = rt_last_kernel_ms();
  if (ms > 0.f) P.mrays_per_s = (f32)((f64)stats.rays / ((f64)ms * 1e3));
  rt_present_upload(P.rgba);
}

void rt_present_panel() {
  auto &P = gRt_Present;
  ImGui::Begin("rt");
  ImGui::Image((ImTextureID)P.view, ImVec2((f32)P.width, (f32)P.height));
  ImGui::End();

  ImGui::Begin("trace settings");
  ImGui::Text("%u samples shown, %.0f Mrays/s (trace kernel)", P.samples_shown, P.mrays_per_s);
  int spp = (int)P.settings.spp, depth = (int)P.settings.max_depth;
  // anything that changes what a sample computes restarts the accumulation; spp per frame does not
  if (ImGui::SliderInt("samples per frame", &spp, 1, 64)) P.settings.spp = (u32)spp;
  if (ImGui::SliderInt("max depth", &depth, 1, 32)) { P.settings.max_depth = (u32)depth; P.restart = true; }
  P.restart |= ImGui::DragFloat3("camera from", P.cam_from.v, 0.05f);
  P.restart |= ImGui::DragFloat3("camera at", P.cam_at.v, 0.05f);
  P.restart |= ImGui::SliderFloat("vertical fov", &P.vfov_deg, 5.f, 120.f);
  P.restart |= ImGui::SliderFloat("aperture", &P.aperture, 0.f, 1.f);
  P.restart |= ImGui::SliderFloat("focus distance", &P.focus_dist, 0.1f, 50.f);
  P.restart |= ImGui::ColorEdit3("sky, horizon", P.settings.sky_bottom);
  P.restart |= ImGui::ColorEdit3("sky, zenith", P.settings.sky_top);
  if (ImGui::Button("restart accumulation")) P.restart = true;
  ImGui::End();
}
}  // namespace rt
