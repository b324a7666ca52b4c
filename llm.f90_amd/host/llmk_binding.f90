! ISO_C_BINDING interface to the llmk C-ABI (include/llmk.h).  One explicit interface per
! exported symbol; constants mirror the header.  This is the whole device boundary of the host:
! the call `logits = transformer(token,pos,s,weights)` (/root/reference/llama2.f90:380) becomes
! `rc = llmk_forward(ctx, token, pos, logits)`.
module llmk_binding
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: LLMK_TYPE_F32 = 0, LLMK_TYPE_F16 = 1, LLMK_TYPE_Q4_0 = 2, LLMK_TYPE_Q6_K = 14
  integer(c_int), parameter :: LLMK_TOKEN_EMBEDDING_TABLE = 0, LLMK_RMS_ATT_WEIGHT = 1, LLMK_RMS_FFN_WEIGHT = 2, &
       LLMK_WQKV = 3, LLMK_WO = 4, LLMK_W13 = 5, LLMK_W2 = 6, LLMK_RMS_FINAL_WEIGHT = 7, LLMK_WCLS = 8
  integer(c_int), parameter :: LLMK_FLAG_NO_GRAPH = 1, LLMK_FLAG_TIMINGS = 2

  type, bind(C) :: llmk_config
     integer(c_int32_t) :: emb_dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len
     integer(c_int32_t) :: weight_type, device, flags
  end type llmk_config

  interface
     integer(c_int) function llmk_create(cfg, ctx) bind(C, name="llmk_create")
       import :: c_int, c_ptr, llmk_config
       type(llmk_config), intent(in) :: cfg
       type(c_ptr), intent(out) :: ctx
     end function
     ! tensor-parallel shard `tp_rank` of `tp_size` (the 70B configuration): llmk_upload is still handed the FULL arrays
     integer(c_int) function llmk_create_tp(cfg, tp_rank, tp_size, ctx) bind(C, name="llmk_create_tp")
       import :: c_int, c_ptr, llmk_config
       type(llmk_config), intent(in) :: cfg
       integer(c_int), value :: tp_rank, tp_size
       type(c_ptr), intent(out) :: ctx
     end function
     integer(c_int) function llmk_tp_unique_id(id_out) bind(C, name="llmk_tp_unique_id")
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id_out(128)
     end function
     integer(c_int) function llmk_tp_init_comm(ctx, id) bind(C, name="llmk_tp_init_comm")
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: id(128)
     end function
     integer(c_int) function llmk_tp_p2p_handle(ctx, handle_out) bind(C, name="llmk_tp_p2p_handle")
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(out) :: handle_out(64)
     end function
     integer(c_int) function llmk_tp_p2p_connect(ctx, handles) bind(C, name="llmk_tp_p2p_connect")
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: handles(*)     ! tp_size * 64 bytes, rank order
     end function
     ! all ranks together, after the connect: 0 = this rank's peer-memory exchanges gave exact sums on this hardware
     integer(c_int) function llmk_tp_p2p_selftest(ctx, iters) bind(C, name="llmk_tp_p2p_selftest")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: iters
     end function
     integer(c_int) function llmk_tp_p2p_disable(ctx) bind(C, name="llmk_tp_p2p_disable")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function llmk_upload(ctx, tensor_id, host, nbytes, ggml_type) bind(C, name="llmk_upload")
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: tensor_id
       type(c_ptr), value :: host
       integer(c_size_t), value :: nbytes
       integer(c_int), value :: ggml_type
     end function
     integer(c_int) function llmk_upload_rows(ctx, tensor_id, layer, row_offset, rows, host, nbytes, ggml_type) &
          bind(C, name="llmk_upload_rows")
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: tensor_id, layer, row_offset, rows
       type(c_ptr), value :: host
       integer(c_size_t), value :: nbytes
       integer(c_int), value :: ggml_type
     end function
     integer(c_int) function llmk_set_tensor_type(ctx, tensor_id, ggml_type) bind(C, name="llmk_set_tensor_type")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: tensor_id, ggml_type
     end function
     integer(c_int) function llmk_set_rms_eps(ctx, eps) bind(C, name="llmk_set_rms_eps")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       real(c_float), value :: eps
     end function
     integer(c_int) function llmk_set_rope_freqs(ctx, freqs, n) bind(C, name="llmk_set_rope_freqs")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       real(c_float), intent(in) :: freqs(*)
       integer(c_int), value :: n
     end function
     integer(c_int) function llmk_forward(ctx, token, pos, logits) bind(C, name="llmk_forward")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       integer(c_int), value :: token, pos
       real(c_float), intent(out) :: logits(*)
     end function
     integer(c_int) function llmk_prefill(ctx, tokens, n, pos0, logits) bind(C, name="llmk_prefill")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       integer(c_int), intent(in) :: tokens(*)
       integer(c_int), value :: n, pos0
       real(c_float), intent(out) :: logits(*)
     end function
     integer(c_int) function llmk_forward_greedy(ctx, token, pos, next_token) bind(C, name="llmk_forward_greedy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: token, pos
       integer(c_int), intent(out) :: next_token
     end function
     ! n positions at temperature 0 with the argmax on the device and no host round trip per token; on_token
     ! (void(int index, int token, void* user), or c_null_funptr) is called in order as the ids arrive
     integer(c_int) function llmk_decode_greedy(ctx, token, pos0, n, ids_out, on_token, user) bind(C, name="llmk_decode_greedy")
       import :: c_int, c_ptr, c_funptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: token, pos0, n
       integer(c_int), intent(out) :: ids_out(*)
       type(c_funptr), value :: on_token
       type(c_ptr), value :: user
     end function
     integer(c_int) function llmk_path(ctx) bind(C, name="llmk_path")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function llmk_tk_shapes(buf, n) bind(C, name="llmk_tk_shapes")
       import :: c_int, c_char, c_size_t
       character(kind=c_char), intent(out) :: buf(*)
       integer(c_size_t), value :: n
     end function
     integer(c_int) function llmk_reset(ctx) bind(C, name="llmk_reset")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function llmk_timings(ctx, ms) bind(C, name="llmk_timings")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: ctx
       real(c_float), intent(out) :: ms(5)
     end function
     integer(c_int) function llmk_destroy(ctx) bind(C, name="llmk_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     type(c_ptr) function llmk_strerror(code) bind(C, name="llmk_strerror")
       import :: c_int, c_ptr
       integer(c_int), value :: code
     end function
     integer(c_int) function llmk_version() bind(C, name="llmk_version")
       import :: c_int
     end function
     ! libc, for the multi-process rendezvous of `llm --ngpu N`
     integer(c_int) function c_getpid() bind(C, name="getpid")
       import :: c_int
     end function
     type(c_ptr) function c_mkdtemp(template) bind(C, name="mkdtemp")       ! creates the directory, mode 0700
       import :: c_ptr, c_char
       character(kind=c_char), intent(inout) :: template(*)
     end function
     integer(c_int) function c_usleep(us) bind(C, name="usleep")
       import :: c_int
       integer(c_int), value :: us
     end function
  end interface

contains

  ! print + stop, the reference's own error convention (read_ggml.f90:122-125)
  subroutine llmk_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    integer :: n
    if (rc == 0) return
    call c_f_pointer(llmk_strerror(rc), msg, [256])
    n = 0
    do while (n < 256)
       if (msg(n+1) == c_null_char) exit
       n = n + 1
    end do
    print *, what, ": ", msg(1:n), " (code", rc, ")"
    stop 1
  end subroutine

end module llmk_binding
