! Host-side data layout of the MI355X-native llm.f90 decode path.
!
! Drop-in surface: module and type names, component names, ranks and index order are the
! reference's (/root/reference/weight_module.f90:2-41), so code written against
! `use weight_module` keeps compiling.  Fortran (in, rows, layer) column-major is exactly the
! C [layer][row][in] the HIP kernels stream, so arrays go to the device with c_loc() and no
! transposition (include/llmk.h, llmk_upload).
!
! Additions (not in the reference): every array component is reachable through a `target`
! variable for c_loc(); `wtype` and the *_raw byte arrays hold f16 (ggml type 1) / q4_0 (type 2)
! matrices exactly as they sit in the GGUF file -- they are never dequantised on the host.

module precision_module
  implicit none
  integer, parameter :: wp = kind(1.0)   ! 4-byte reals, as weight_module.f90:4
end module precision_module

module weight_module
  use iso_c_binding, only: c_int8_t
  use precision_module
  implicit none
  private wp

  type TransformerWeights
     ! f32 path (reference layout, weight_module.f90:13-26)
     real(kind=wp), allocatable :: token_embedding_table(:,:)   ! (emb, vocab)
     real(kind=wp), allocatable :: rms_att_weight(:,:)          ! (emb, layer)
     real(kind=wp), allocatable :: rms_ffn_weight(:,:)          ! (emb, layer)
     real(kind=wp), allocatable :: wqkv(:,:,:)                  ! (emb, emb+2*kv, layer): Q | K | V rows
     real(kind=wp), allocatable :: wo(:,:,:)                    ! (emb, emb, layer)
     real(kind=wp), allocatable :: w13(:,:,:)                   ! (emb, 2*hidden, layer): gate | up rows
     real(kind=wp), allocatable :: w2(:,:,:)                    ! (hidden, emb, layer)
     real(kind=wp), allocatable :: rms_final_weight(:)          ! (emb)
     real(kind=wp), allocatable :: wcls(:,:)                    ! (emb, vocab)
     ! f16 / q4_0 path: ggml type of the five matrices and their bytes in the same fused order
     integer :: wtype = 0
     integer :: wcls_type = 0      ! ggml type the classifier is handed over in: the matrices', or 14 (raw q6_K super-blocks: a stock q4_0 file's
                                   ! output.weight, dotted on the device), or f32 when the loader dequantised a foreign type
     integer(c_int8_t), allocatable :: wqkv_raw(:), wo_raw(:), w13_raw(:), w2_raw(:), wcls_raw(:)
  end type TransformerWeights

  type Config
     integer :: emb_dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len
     integer :: kv_head_size
     ! read from the file, used only on request (llm --gguf-eps / --gguf-rope-base); 0 = key absent
     real(kind=wp) :: rms_eps = 0, rope_freq_base = 0
  end type Config

  ! The KV cache and attention scratch live on the device inside the llmk context; the host
  ! keeps the reference's RunState only for its timers (and so that `type(RunState)` exists).
  type RunState
     real(kind=wp), allocatable :: att(:,:)
     real(kind=wp), allocatable :: key_cache(:,:,:)
     real(kind=wp), allocatable :: value_cache(:,:,:)
     real(kind=wp) :: times(5)
  end type RunState

end module weight_module
