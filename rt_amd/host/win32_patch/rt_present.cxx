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
  rt_panel_init(P.panel, width, height);
  P.rgba = (u32 *)alloc_perm((s64)width * height * (s64)sizeof(u32));

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
  u32 const width = P.panel.width, height = P.panel.height;
  for (u32 y = 0; y < height; y++) {
    u32 const *src = rgba8 + (s64)(height - 1 - y) * width;              // frame row 0 is the bottom row
    u8        *dst = (u8 *)mapped.pData + (s64)y * mapped.RowPitch;
    for (u32 x = 0; x < width; x++) {
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
  rt_panel_step_or_panic(P.panel, P.rgba);   // the whole state machine lives in rt.cpp (tested on Linux)
  rt_present_upload(P.rgba);
}

// The widgets only fill an Rt_Panel_Edit; what restarts the accumulation is rt_panel_apply's business.
void rt_present_panel() {
  auto &P = gRt_Present;
  auto &S = P.panel;
  ImGui::Begin("rt");
  ImGui::Image((ImTextureID)P.view, ImVec2((f32)S.width, (f32)S.height));
  ImGui::End();

  ImGui::Begin("trace settings");
  ImGui::Text("%u samples shown, %.0f Mrays/s (trace kernel)", S.samples_shown, S.mrays_per_s);
  Rt_Panel_Edit edit = {};
  int spp = (int)S.settings.spp, depth = (int)S.settings.max_depth;
  if (ImGui::SliderInt("samples per frame", &spp, 1, 64)) { edit.set_spp = true; edit.spp = (u32)spp; }
  if (ImGui::SliderInt("max depth", &depth, 1, 32)) { edit.set_depth = true; edit.max_depth = (u32)depth; }
  edit.cam_from = S.cam_from; edit.cam_at = S.cam_at;
  edit.vfov_deg = S.vfov_deg; edit.aperture = S.aperture; edit.focus_dist = S.focus_dist;
  edit.set_camera |= ImGui::DragFloat3("camera from", edit.cam_from.v, 0.05f);
  edit.set_camera |= ImGui::DragFloat3("camera at", edit.cam_at.v, 0.05f);
  edit.set_camera |= ImGui::SliderFloat("vertical fov", &edit.vfov_deg, 5.f, 120.f);
  edit.set_camera |= ImGui::SliderFloat("aperture", &edit.aperture, 0.f, 1.f);
  edit.set_camera |= ImGui::SliderFloat("focus distance", &edit.focus_dist, 0.1f, 50.f);
  for (int k = 0; k < 3; k++) { edit.sky_bottom[k] = S.settings.sky_bottom[k]; edit.sky_top[k] = S.settings.sky_top[k]; }
  edit.set_sky |= ImGui::ColorEdit3("sky, horizon", edit.sky_bottom);
  edit.set_sky |= ImGui::ColorEdit3("sky, zenith", edit.sky_top);
  edit.restart_pressed = ImGui::Button("restart accumulation");
  ImGui::End();
  rt_panel_apply(S, edit);
}
}  // namespace rt
