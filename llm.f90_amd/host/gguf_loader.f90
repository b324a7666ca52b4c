! GGUF model loader for the Fortran host.  Drop-in for the reference's
!   call load_ggml(filename, w, c, vocab, scores, token_lengths, v)     (/root/reference/read_ggml.f90:53-60)
! same module name, same argument list, same fused weight layout (Q|K|V rows in wqkv, gate|up
! rows in w13: read_ggml.f90:272,286,300,347,376), same console output in the non-verbose case
! (the unconditional " data offset" line, read_ggml.f90:196).
!
! Written from the GGUF file format, not from the reference's code.  Differences on purpose:
!   * every GGUF metadata value type is understood (the reference stops on anything but
!     int32/uint32/float32/string/array, read_ggml.f90:663-685) so stock llama.cpp files load;
!   * matrices of ggml type 1 (f16) and 2 (q4_0) are kept as raw bytes (w%*_raw, w%wtype) and
!     go to the GPU undecoded; the embedding table (gathered, not streamed) is always widened
!     to f32 on the host;
!   * positioning uses standard stream I/O (read(..., pos=)) instead of the GNU fseek extension.
module read_ggml
  use iso_c_binding
  use precision_module
  use weight_module
  implicit none
  private
  public :: load_ggml, stream_ggml_weights, half_bits_to_real, tensor_sink, verbose_ext
  public :: TID_TOKEN_EMBEDDING_TABLE, TID_RMS_ATT_WEIGHT, TID_RMS_FFN_WEIGHT, TID_WQKV, TID_WO, TID_W13, TID_W2, &
            TID_RMS_FINAL_WEIGHT, TID_WCLS

  ! Where stream_ggml_weights hands its tensors: rows [row_offset, row_offset + rows) of layer `layer` (0-based) of the
  ! fused TransformerWeights component `tensor_id` (weight_module.f90:13-26; the TID_* numbers are include/llmk.h's
  ! LLMK_* tensor ids), `nbytes` bytes in ggml type `ggml_type` at `host`.  The host's sink calls llmk_upload_rows; the
  ! loader itself knows nothing about the device (tests/host_tools/loader_dump.f90 links it without the library).
  abstract interface
     subroutine tensor_sink(tensor_id, layer, row_offset, rows, host, nbytes, ggml_type)
       import :: c_ptr, c_size_t
       integer, intent(in) :: tensor_id, layer, row_offset, rows, ggml_type
       type(c_ptr), intent(in) :: host
       integer(c_size_t), intent(in) :: nbytes
     end subroutine
  end interface
  integer, parameter :: TID_TOKEN_EMBEDDING_TABLE = 0, TID_RMS_ATT_WEIGHT = 1, TID_RMS_FFN_WEIGHT = 2, TID_WQKV = 3, TID_WO = 4, &
                        TID_W13 = 5, TID_W2 = 6, TID_RMS_FINAL_WEIGHT = 7, TID_WCLS = 8

  integer(4), parameter :: GGUF_MAGIC = 1179993927     ! "GGUF", read_ggml.f90:122
  integer, parameter :: NAME_LEN = 64                  ! reference truncates names/tokens to 64 chars
  integer, parameter :: GT_F32 = 0, GT_F16 = 1, GT_Q4_0 = 2, GT_Q6_K = 14
  real(8) :: last_real = 0        ! value of the last float-typed scalar read_scalar saw (it returns integers)

  type :: tensor_entry
     character(len=NAME_LEN) :: name = ""
     integer :: ndim = 0
     integer(8) :: dims(4) = 1
     integer :: ttype = 0
     integer(8) :: offset = 0
  end type tensor_entry

  type(tensor_entry), allocatable :: dir(:)
  integer(8) :: data_pos          ! 1-based stream position of the tensor data section
  integer :: u                    ! unit
  logical :: deferred = .false.   ! load_ggml(..., defer=.true.) left the matrices in the (still open) file
  ! `-v` prints the reference's lines, in the reference's order and format (read_ggml.f90:114-120, 136, 155-156, 181-182,
  ! 208-214, 234-235, 245-413, 438, 457, 467, 479); what this loader can say beyond them (matrix type, streaming) is printed
  ! only when the host also sets this (llm --vx)
  logical :: verbose_ext = .false.

contains

  ! defer (optional, an extension; absent or .false. = the reference's behaviour): read everything BUT the embedding
  ! table and the five matrix families, and keep the file open -- stream_ggml_weights then moves them to the device one
  ! tensor (or row range) at a time, so the host never holds more than the largest single tensor it hands over
  ! (`llm --ngpu N`: each rank reads only the rows of its own shard).
  subroutine load_ggml(filename, w, c, vocab, scores, token_lengths, v, defer)
    character(len=*), intent(in) :: filename
    type(TransformerWeights), intent(out) :: w
    type(Config), intent(out) :: c
    real(kind=wp), allocatable, intent(out) :: scores(:)
    character(:), dimension(:), allocatable, intent(out) :: vocab
    integer(4), allocatable, intent(out) :: token_lengths(:)
    logical, intent(in) :: v
    logical, intent(in), optional :: defer

    integer(4) :: magic, version, vtype, etype
    integer(8) :: n_tensors, n_kv, i, j, n, slen, p, tokens_pos, n_tokens, ival
    integer :: alignment, ios, width, l
    integer(4) :: n_layers, emb, ctx_len, n_heads, n_kv_heads, ffn, vocab_size
    character(len=:), allocatable :: key, str
    logical :: have_kv_heads
    integer :: E, H, KV, hs, mt, cls_ft, envstat
    character(len=8) :: envbuf
    integer(1) :: b3(3)
    character(len=NAME_LEN) :: tname

    alignment = 32; n_layers = 0; emb = 0; ctx_len = 0; n_heads = 0; n_kv_heads = 0; ffn = 0
    vocab_size = 0; tokens_pos = 0; n_tokens = 0; have_kv_heads = .false.

    open(newunit=u, file=filename, form="unformatted", access="stream", status="old", action="read", iostat=ios)
    if (ios /= 0) then
       print *, "cannot open model file ", trim(filename)
       stop 1
    end if

    read(u) magic, version, n_tensors, n_kv
    if (v) then
       print *, "GGUF Header Info"
       print *, "Magic number: ", magic
       print *, "Version: ", version
       print *, "Tensor Count: ", n_tensors
       print *, "Key-Value Pairs: ", n_kv
    end if
    if (magic /= GGUF_MAGIC) then
       print *, "Magic numbers do not match, exiting"
       stop
    end if

    ! ---- metadata ---------------------------------------------------------------------------
    do i = 1, n_kv
       call read_string(key)
       read(u) vtype
       select case (vtype)
       case (8)
          call read_string(str)
          if (v) then
             print *, pad64(key)                ! print *, keys(i): character(len=64)          read_ggml.f90:155
             print *, pad64(str)                ! print_multi: m%string is character(64)       :692-693
          end if
       case (9)
          read(u) etype, n
          if (key == "tokenizer.ggml.tokens" .and. etype == 8) then
             inquire(unit=u, pos=tokens_pos)
             n_tokens = n
             vocab_size = int(n, 4)
             do j = 1, n                      ! first pass: only skip (lengths are re-read below)
                read(u) slen
                inquire(unit=u, pos=p)
                read(u, pos=p + slen - 1) b3(1)
             end do
          else if (key == "tokenizer.ggml.scores" .and. etype == 6) then
             allocate(scores(n))
             read(u) scores
          else
             call skip_array(etype, n)
          end if
          if (v) then
             print *, pad64(key)
             print *, int(n, 4)                 ! print *, size(m%a)                           :700-701
          end if
       case default
          ival = read_scalar(vtype)
          if (v) then
             if (key == "general.alignment") print *, "alignment set to", int(ival, 4)          ! :136 (before the key line)
             print *, pad64(key)
             select case (vtype)
             case (4, 5); print *, int(ival, 4)                                                  ! m%i32   :694-697
             case (6);    print *, real(last_real, 4)                                            ! m%f32   :698-699
             case (12);   print *, last_real      ! (the types below stop the reference: "Not implemented", :682-684)
             case default; print *, ival
             end select
          end if
          select case (key)
          case ("general.alignment");               alignment = int(ival)
          case ("llama.block_count");                n_layers = int(ival, 4)
          case ("llama.embedding_length");           emb = int(ival, 4)
          case ("llama.attention.head_count");       n_heads = int(ival, 4)
          case ("llama.attention.head_count_kv");    n_kv_heads = int(ival, 4); have_kv_heads = .true.
          case ("llama.context_length");             ctx_len = int(ival, 4)
          case ("llama.feed_forward_length");        ffn = int(ival, 4)
          ! kept for hosts that opt in (llm --gguf-eps / --gguf-rope-base); the reference ignores both (llama2.f90:454,545)
          case ("llama.attention.layer_norm_rms_epsilon"); c%rms_eps = real(last_real, wp)
          case ("llama.rope.freq_base");                   c%rope_freq_base = real(last_real, wp)
          end select
       end select
    end do
    if (.not. have_kv_heads) n_kv_heads = n_heads
    if (n_layers <= 0 .or. emb <= 0 .or. n_heads <= 0 .or. ffn <= 0 .or. vocab_size <= 0) then
       print *, "model file lacks llama.* shape keys or a tokenizer"
       stop 1
    end if

    ! ---- tensor directory -------------------------------------------------------------------
    allocate(dir(n_tensors))
    do i = 1, n_tensors
       call read_string(str)
       dir(i)%name = str
       read(u) dir(i)%ndim
       if (dir(i)%ndim > 4) then
          print *, "Ndims not supported", dir(i)%ndim
          stop 1
       end if
       do j = 1, dir(i)%ndim
          read(u) dir(i)%dims(j)
       end do
       read(u) dir(i)%ttype
       read(u) dir(i)%offset
    end do

    ! data section starts at the next multiple of `alignment` (0-based offset), read_ggml.f90:176-194
    inquire(unit=u, pos=p)
    data_pos = ((p - 1 + alignment - 1) / alignment) * alignment + 1
    if (v) then
       print *, "Position", p
       print *, "Deficit", mod(p - 1, int(alignment, 8))
    end if
    print *, "data offset", int(data_pos, 4)

    E = emb; H = ffn; hs = emb / n_heads; KV = n_kv_heads * hs
    c%emb_dim = emb; c%hidden_dim = ffn; c%n_layers = n_layers; c%n_heads = n_heads
    c%n_kv_heads = n_kv_heads; c%vocab_size = vocab_size; c%seq_len = ctx_len; c%kv_head_size = KV
    if (v) then
       print *, "Embedding dimension: ", emb
       print *, "Hidden dimension: ", ffn
       print *, "Layers: ", n_layers
       print *, "Heads: ", n_heads
       print *, "kv Heads: ", n_kv_heads
       print *, "Vocabulary Size: ", vocab_size
       print *, "Sequence Length: ", ctx_len
       print *, "head size ", hs
       print *, "kv head Size ", KV
    end if

    ! ---- weights ----------------------------------------------------------------------------
    mt = dir(find("blk.0.attn_q.weight"))%ttype
    if (mt /= GT_F32 .and. mt /= GT_F16 .and. mt /= GT_Q4_0) then
       print *, "Type not supported", mt
       stop 1
    end if
    w%wtype = mt
    w%wcls_type = mt
    deferred = .false.
    if (present(defer)) deferred = defer
    ! stock llama.cpp q4_0 files keep output.weight in q6_K (the reference stops on any type >= 2, read_ggml.f90:633-635).
    ! Round 6: its super-blocks go to the device AS THEY LIE IN THE FILE (llmk.h LLMK_TYPE_Q6_K; rows are whole super-blocks
    ! when emb_dim is a multiple of 256) and the q4_0 persistent kernels dot them there; any other foreign type -- or
    ! LLM_DEQUANT_CLS=1 in the environment, the round-2 behaviour -- is dequantised here exactly as ggml does and handed over as f32.
    cls_ft = dir(find("output.weight"))%ttype
    if (cls_ft /= mt) then
       call get_environment_variable("LLM_DEQUANT_CLS", envbuf, status=envstat)
       if (cls_ft == GT_Q6_K .and. mt /= GT_F32 .and. mod(E, 256) == 0 .and. .not. (envstat == 0 .and. envbuf(1:1) == "1")) then
          w%wcls_type = GT_Q6_K
       else
          w%wcls_type = GT_F32
       end if
    end if

    if (.not. deferred) then
       allocate(w%token_embedding_table(E, vocab_size))
       call read_matrix_as_f32("token_embd.weight", w%token_embedding_table, E, vocab_size)
    end if

    allocate(w%rms_att_weight(E, n_layers), w%rms_ffn_weight(E, n_layers), w%rms_final_weight(E))
    do l = 1, n_layers
       call read_vector(layer_name(l, "attn_norm.weight"), w%rms_att_weight(:, l), E)
       call read_vector(layer_name(l, "ffn_norm.weight"), w%rms_ffn_weight(:, l), E)
    end do
    call read_vector("output_norm.weight", w%rms_final_weight, E)

    if (deferred) then
       if (v .and. verbose_ext) print *, "matrices stay in the file (streamed to the device), ggml type", mt
    else if (mt == GT_F32) then
       allocate(w%wqkv(E, E + 2*KV, n_layers), w%wo(E, E, n_layers), w%w13(E, 2*H, n_layers), &
                w%w2(H, E, n_layers), w%wcls(E, vocab_size))
       do l = 1, n_layers
          call read_f32(layer_name(l, "attn_q.weight"), w%wqkv(:, 1:E, l), E, E)
          call read_f32(layer_name(l, "attn_k.weight"), w%wqkv(:, E+1:E+KV, l), E, KV)
          call read_f32(layer_name(l, "attn_v.weight"), w%wqkv(:, E+KV+1:E+2*KV, l), E, KV)
          call read_f32(layer_name(l, "attn_output.weight"), w%wo(:, :, l), E, E)
          call read_f32(layer_name(l, "ffn_gate.weight"), w%w13(:, 1:H, l), E, H)
          call read_f32(layer_name(l, "ffn_up.weight"), w%w13(:, H+1:2*H, l), E, H)
          call read_f32(layer_name(l, "ffn_down.weight"), w%w2(:, :, l), H, E)
       end do
       if (cls_ft == GT_F32) then
          call read_f32("output.weight", w%wcls, E, vocab_size)
       else
          call read_matrix_as_f32("output.weight", w%wcls, E, vocab_size)
       end if
    else
       allocate(w%wqkv_raw(rowbytes(mt, E) * (E + 2*KV) * n_layers), w%wo_raw(rowbytes(mt, E) * E * n_layers), &
                w%w13_raw(rowbytes(mt, E) * 2*H * n_layers), w%w2_raw(rowbytes(mt, H) * E * n_layers), &
                w%wcls_raw(rowbytes(mt, E) * vocab_size))
       do l = 1, n_layers
          call read_raw(layer_name(l, "attn_q.weight"), w%wqkv_raw, mt, E, E, int(l-1, 8) * (E + 2*KV))
          call read_raw(layer_name(l, "attn_k.weight"), w%wqkv_raw, mt, E, KV, int(l-1, 8) * (E + 2*KV) + E)
          call read_raw(layer_name(l, "attn_v.weight"), w%wqkv_raw, mt, E, KV, int(l-1, 8) * (E + 2*KV) + E + KV)
          call read_raw(layer_name(l, "attn_output.weight"), w%wo_raw, mt, E, E, int(l-1, 8) * E)
          call read_raw(layer_name(l, "ffn_gate.weight"), w%w13_raw, mt, E, H, int(l-1, 8) * 2*H)
          call read_raw(layer_name(l, "ffn_up.weight"), w%w13_raw, mt, E, H, int(l-1, 8) * 2*H + H)
          call read_raw(layer_name(l, "ffn_down.weight"), w%w2_raw, mt, H, E, int(l-1, 8) * E)
       end do
       if (w%wcls_type == mt) then
          call read_raw("output.weight", w%wcls_raw, mt, E, vocab_size, 0_8)
       else if (w%wcls_type == GT_Q6_K) then      ! raw super-blocks (see above)
          deallocate(w%wcls_raw)
          allocate(w%wcls_raw(rowbytes(GT_Q6_K, E) * vocab_size))
          call read_raw("output.weight", w%wcls_raw, GT_Q6_K, E, vocab_size, 0_8)
       else
          deallocate(w%wcls_raw)
          allocate(w%wcls(E, vocab_size))
          call read_matrix_as_f32("output.weight", w%wcls, E, vocab_size)
          w%wcls_type = GT_F32
       end if
    end if
    if (v) then
       ! the reference's twelve lines in its order (read_ggml.f90:245-413: it loads family by family, this loader layer by
       ! layer -- the counts are the arrays' element counts whatever the matrices' encoding, and whether or not they are
       ! still in the file)
       call loaded("loaded embedding weights:", int(E, 8) * vocab_size)
       call loaded("loaded rms att weights:", int(E, 8) * n_layers)
       call loaded("loaded wq weights:", int(E, 8) * E * n_layers)
       call loaded("loaded wk weights:", int(E, 8) * KV * n_layers)
       call loaded("loaded wv weights:", int(E, 8) * KV * n_layers)
       call loaded("loaded wo weights:", int(E, 8) * E * n_layers)
       call loaded("loaded ffn norm weights:", int(E, 8) * n_layers)
       call loaded("loaded w1 (gate) weights:", int(E, 8) * H * n_layers)
       call loaded("loaded w2 (down) weights:", int(E, 8) * H * n_layers)
       call loaded("loaded w3 (up) weights:", int(E, 8) * H * n_layers)
       call loaded("loaded output norm weights:", int(E, 8))
       call loaded("loaded classifier weights:", int(E, 8) * vocab_size)
       if (verbose_ext .and. .not. deferred) print *, "loaded matmul weights, ggml type", mt
    end if

    ! ---- vocabulary (second visit of the tokens array) ---------------------------------------
    if (tokens_pos == 0 .or. .not. allocated(scores)) then
       print *, "model file has no tokenizer.ggml.tokens / tokenizer.ggml.scores"
       stop 1
    end if
    width = NAME_LEN
    allocate(character(len=width) :: vocab(n_tokens))
    allocate(token_lengths(n_tokens))
    p = tokens_pos
    do j = 1, n_tokens
       read(u, pos=p) slen
       allocate(character(len=int(slen)) :: str)
       if (slen > 0) read(u) str
       p = p + 8 + slen
       n = min(slen, int(width, 8))
       vocab(j) = str(1:n)
       token_lengths(j) = int(n, 4)
       ! sentencepiece's U+2581 (bytes E2 96 81) in front of a token stands for a space (read_ggml.f90:479-497)
       if (n >= 3) then
          if (iachar(str(1:1)) == 226 .and. iachar(str(2:2)) == 150 .and. iachar(str(3:3)) == 129) then
             vocab(j) = " " // str(4:n)
             token_lengths(j) = int(n - 2, 4)
          end if
       end if
       deallocate(str)
    end do
    if (v) then
       print *, "loading tokens"                                                                  ! read_ggml.f90:438
       write (*, "(A,I0,A)") "found ", size(vocab), " tokens"
       write (*, "(A,I0,A)") "found ", size(scores), " scores"
       print *, "maximum token length ", maxval(token_lengths)
    end if

    if (.not. deferred) then
       close(u)
       deallocate(dir)
    end if

  contains
    ! `print *, "loaded ...", size(array)`: a default integer in the reference (element counts beyond it wrap there)
    subroutine loaded(what, n)
      character(len=*), intent(in) :: what
      integer(8), intent(in) :: n
      if (n <= huge(1_4)) then
         print *, what, int(n, 4)
      else
         print *, what, n
      end if
    end subroutine
  end subroutine load_ggml

  ! a string as the reference holds it: character(len=64), blank-padded or cut (read_ggml.f90:15, 66, 127)
  function pad64(sv) result(r)
    character(len=*), intent(in) :: sv
    character(len=NAME_LEN) :: r
    r = sv
  end function

  function layer_name(l1, suffix) result(nm)
    integer, intent(in) :: l1
    character(len=*), intent(in) :: suffix
    character(len=NAME_LEN) :: nm
    write (nm, "(A,I0,A,A)") "blk.", l1 - 1, ".", suffix
  end function

  ! Second half of a deferred load: the embedding table, the norm gains (already in w) and the matrices go to the llmk
  ! context `ctx` -- tensor by tensor through llmk_upload_rows, whose row numbers are those of the FULL fused tensors
  ! (wqkv = Q | K | V rows, w13 = gate | up rows: read_ggml.f90:272-376).  Rank tp_rank of tp_size reads from the file only
  ! what its shard keeps: its query heads' rows of attn_q, its kv heads' rows of attn_k / attn_v, its hidden rows of
  ! ffn_gate / ffn_up, its vocabulary rows of output.weight; attn_output and ffn_down are split along the contraction, so
  ! their rows are read whole and the shim keeps the column slice.  The largest buffer is one ffn_down tensor (or 4096
  ! embedding rows).  Replaces the array assignments of read_ggml.f90:238-410 for hosts that opt in.
  subroutine stream_ggml_weights(sink, w, c, tp_rank, tp_size, v)
    procedure(tensor_sink) :: sink
    type(TransformerWeights), intent(inout), target :: w
    type(Config), intent(in) :: c
    integer, intent(in) :: tp_rank, tp_size
    logical, intent(in) :: v
    integer :: E, H, KV, hs, L, V_, mt, l1, Eq, KVl, Hl, Vl, r0, nr, chunk
    integer(c_int8_t), allocatable, target :: raw(:)
    real(kind=wp), allocatable, target :: f32buf(:, :)
    integer(8) :: peak

    if (.not. deferred) then
       print *, "stream_ggml_weights: load_ggml was not called with defer=.true."
       stop 1
    end if
    E = c%emb_dim; H = c%hidden_dim; L = c%n_layers; V_ = c%vocab_size
    hs = E / c%n_heads; KV = c%n_kv_heads * hs; mt = w%wtype
    Eq = E / tp_size; KVl = KV / tp_size; Hl = H / tp_size; Vl = V_ / tp_size
    peak = 0

    ! replicated f32 tensors: the embedding table in chunks of rows, the gains from w
    chunk = min(V_, 4096)
    allocate(f32buf(E, chunk))
    r0 = 0
    do while (r0 < V_)
       nr = min(chunk, V_ - r0)
       call read_rows_as_f32("token_embd.weight", f32buf, E, V_, r0, nr)
       call sink(TID_TOKEN_EMBEDDING_TABLE, 0, r0, nr, c_loc(f32buf), 4_c_size_t * E * nr, GT_F32)
       r0 = r0 + nr
    end do
    do l1 = 1, L
       call sink(TID_RMS_ATT_WEIGHT, l1 - 1, 0, 1, c_loc(w%rms_att_weight(1, l1)), 4_c_size_t * E, GT_F32)
       call sink(TID_RMS_FFN_WEIGHT, l1 - 1, 0, 1, c_loc(w%rms_ffn_weight(1, l1)), 4_c_size_t * E, GT_F32)
    end do
    call sink(TID_RMS_FINAL_WEIGHT, 0, 0, 1, c_loc(w%rms_final_weight), 4_c_size_t * E, GT_F32)

    ! classifier: this rank's vocabulary rows; dequantised in chunks when the file keeps it in another type (q6_K)
    if (w%wcls_type == mt .or. w%wcls_type == GT_Q6_K) then
       call stream_rows("output.weight", TID_WCLS, 0, E, V_, tp_rank * Vl, Vl, tp_rank * Vl, w%wcls_type)
    else
       r0 = tp_rank * Vl
       do while (r0 < (tp_rank + 1) * Vl)
          nr = min(chunk, (tp_rank + 1) * Vl - r0)
          call read_rows_as_f32("output.weight", f32buf, E, V_, r0, nr)
          call sink(TID_WCLS, 0, r0, nr, c_loc(f32buf), 4_c_size_t * E * nr, GT_F32)
          r0 = r0 + nr
       end do
    end if
    deallocate(f32buf)

    do l1 = 1, L
       call stream_rows(layer_name(l1, "attn_q.weight"), TID_WQKV, l1 - 1, E, E, tp_rank * Eq, Eq, tp_rank * Eq)
       call stream_rows(layer_name(l1, "attn_k.weight"), TID_WQKV, l1 - 1, E, KV, tp_rank * KVl, KVl, E + tp_rank * KVl)
       call stream_rows(layer_name(l1, "attn_v.weight"), TID_WQKV, l1 - 1, E, KV, tp_rank * KVl, KVl, E + KV + tp_rank * KVl)
       call stream_rows(layer_name(l1, "attn_output.weight"), TID_WO, l1 - 1, E, E, 0, E, 0)
       call stream_rows(layer_name(l1, "ffn_gate.weight"), TID_W13, l1 - 1, E, H, tp_rank * Hl, Hl, tp_rank * Hl)
       call stream_rows(layer_name(l1, "ffn_up.weight"), TID_W13, l1 - 1, E, H, tp_rank * Hl, Hl, H + tp_rank * Hl)
       call stream_rows(layer_name(l1, "ffn_down.weight"), TID_W2, l1 - 1, H, E, 0, E, 0)
    end do
    if (v .and. verbose_ext) print *, "streamed matmul weights, ggml type", mt, " largest host buffer (bytes)", peak
    close(u)
    deallocate(dir)
    deferred = .false.

  contains

    ! rows [frow, frow + n) of file tensor `name` (cols x rows_total) -> rows grow.. of layer `layer0` of fused tensor `tid`
    subroutine stream_rows(name, tid, layer0, cols, rows_total, frow, n, grow, ttype)
      character(len=*), intent(in) :: name
      integer, intent(in) :: tid, layer0, cols, rows_total, frow, n, grow
      integer, intent(in), optional :: ttype       ! the tensor's own type when it is not the matrices' (q6_K classifier rows)
      integer :: idx, tt
      integer(8) :: nb
      tt = mt
      if (present(ttype)) tt = ttype
      idx = find(name)
      call check_shape(idx, cols, rows_total)
      if (dir(idx)%ttype /= tt) then
         print *, "Type not supported", dir(idx)%ttype, " (mixed matrix types) for ", trim(name)
         stop 1
      end if
      nb = n * rowbytes(tt, cols)
      if (allocated(raw)) then
         if (size(raw, kind=8) < nb) deallocate(raw)
      end if
      if (.not. allocated(raw)) allocate(raw(nb))
      peak = max(peak, nb)
      read(u, pos=data_pos + dir(idx)%offset + frow * rowbytes(tt, cols)) raw(1:nb)
      call sink(tid, layer0, grow, n, c_loc(raw), int(nb, c_size_t), tt)
    end subroutine

  end subroutine stream_ggml_weights

  ! ---------------------------------------------------------------------------------------------
  subroutine read_string(s)
    character(len=:), allocatable, intent(out) :: s
    integer(8) :: n
    read(u) n
    allocate(character(len=int(n)) :: s)
    if (n > 0) read(u) s
  end subroutine

  ! any non-string, non-array GGUF value as a 64-bit integer (floats are truncated; unused)
  function read_scalar(vtype) result(val)
    integer(4), intent(in) :: vtype
    integer(8) :: val
    integer(1) :: i1
    integer(2) :: i2
    integer(4) :: i4
    integer(8) :: i8
    real(4) :: r4
    real(8) :: r8
    select case (vtype)
    case (0, 1, 7)                 ! uint8, int8, bool
       read(u) i1; val = iand(int(i1, 8), 255_8)
    case (2, 3)                    ! uint16, int16
       read(u) i2; val = iand(int(i2, 8), 65535_8)
    case (4, 5)                    ! uint32, int32
       read(u) i4; val = int(i4, 8)
    case (6)
       read(u) r4; val = int(r4, 8); last_real = real(r4, 8)
    case (10, 11)
       read(u) i8; val = i8
    case (12)
       read(u) r8; val = int(r8, 8); last_real = r8
    case default
       print *, "Not implemented", vtype          ! the reference's message, read_ggml.f90:683
       stop
    end select
  end function

  subroutine skip_array(etype, n)
    integer(4), intent(in) :: etype
    integer(8), intent(in) :: n
    integer(8) :: j, slen, p, esz
    integer(4) :: sub_t
    integer(8) :: sub_n
    integer(1) :: b
    select case (etype)
    case (0, 1, 7);  esz = 1
    case (2, 3);     esz = 2
    case (4, 5, 6);  esz = 4
    case (10, 11, 12); esz = 8
    case (8)
       do j = 1, n
          read(u) slen
          if (slen > 0) then
             inquire(unit=u, pos=p)
             read(u, pos=p + slen - 1) b
          end if
       end do
       return
    case (9)
       do j = 1, n
          read(u) sub_t, sub_n
          call skip_array(sub_t, sub_n)
       end do
       return
    case default
       print *, "Not implemented", etype
       stop
    end select
    if (n * esz > 0) then
       inquire(unit=u, pos=p)
       read(u, pos=p + n * esz - 1) b
    end if
  end subroutine

  function find(name) result(idx)
    character(len=*), intent(in) :: name
    integer :: idx
    do idx = 1, size(dir)
       if (dir(idx)%name == name) return
    end do
    print *, "key not found", name                 ! read_ggml.f90:571
    stop
  end function

  pure function rowbytes(t, k) result(nb)
    integer, intent(in) :: t, k
    integer(8) :: nb
    select case (t)
    case (GT_F16);  nb = 2_8 * k
    case (GT_Q4_0); nb = int(k / 32, 8) * 18_8
    case (GT_Q6_K); nb = int(k / 256, 8) * 210_8
    case default;   nb = 4_8 * k
    end select
  end function

  subroutine check_shape(idx, cols, rows)
    integer, intent(in) :: idx, cols, rows
    if (dir(idx)%dims(1) /= cols .or. product(dir(idx)%dims(2:max(2, dir(idx)%ndim))) /= rows) then
       print *, "unexpected tensor shape for ", trim(dir(idx)%name), dir(idx)%dims(1:dir(idx)%ndim)
       stop 1
    end if
  end subroutine

  subroutine read_vector(name, dst, n)
    character(len=*), intent(in) :: name
    integer, intent(in) :: n
    real(kind=wp), intent(out) :: dst(n)
    integer :: idx
    idx = find(name)
    if (dir(idx)%ttype /= GT_F32 .or. dir(idx)%dims(1) /= n) then
       print *, "Type not supported", dir(idx)%ttype, " for ", trim(name)
       stop 1
    end if
    read(u, pos=data_pos + dir(idx)%offset) dst
  end subroutine

  subroutine read_f32(name, dst, cols, rows)
    character(len=*), intent(in) :: name
    integer, intent(in) :: cols, rows
    real(kind=wp), intent(out) :: dst(cols, rows)
    integer :: idx
    idx = find(name)
    call check_shape(idx, cols, rows)
    if (dir(idx)%ttype /= GT_F32) then
       print *, "Type not supported", dir(idx)%ttype, " (mixed matrix types) for ", trim(name)
       stop 1
    end if
    read(u, pos=data_pos + dir(idx)%offset) dst
  end subroutine

  subroutine read_raw(name, dst, t, cols, rows, first_row)
    character(len=*), intent(in) :: name
    integer(c_int8_t), intent(inout) :: dst(:)
    integer, intent(in) :: t, cols, rows
    integer(8), intent(in) :: first_row
    integer :: idx
    integer(8) :: b0, nb
    idx = find(name)
    call check_shape(idx, cols, rows)
    if (dir(idx)%ttype /= t) then
       print *, "Type not supported", dir(idx)%ttype, " (mixed matrix types) for ", trim(name)
       stop 1
    end if
    b0 = first_row * rowbytes(t, cols)
    nb = rows * rowbytes(t, cols)
    read(u, pos=data_pos + dir(idx)%offset) dst(b0 + 1:b0 + nb)
  end subroutine

  ! the embedding table is gathered on the device as f32 whatever its file type
  subroutine read_matrix_as_f32(name, dst, cols, rows)
    character(len=*), intent(in) :: name
    integer, intent(in) :: cols, rows
    real(kind=wp), intent(out) :: dst(cols, rows)
    call read_rows_as_f32(name, dst, cols, rows, 0, rows)
  end subroutine

  ! rows [row0, row0 + nrows) of a cols x rows_total file tensor of any supported type, widened to f32
  subroutine read_rows_as_f32(name, dst, cols, rows_total, row0, nrows)
    character(len=*), intent(in) :: name
    integer, intent(in) :: cols, rows_total, row0, nrows
    real(kind=wp), intent(out) :: dst(cols, *)
    integer :: idx, r, b, k, sc, rows
    integer(2), allocatable :: hrow(:)
    integer(1), allocatable :: qrow(:)
    real(kind=wp) :: d
    integer :: q
    integer(8) :: first
    rows = nrows
    idx = find(name)
    call check_shape(idx, cols, rows_total)
    select case (dir(idx)%ttype)
    case (GT_F32)
       first = data_pos + dir(idx)%offset + int(row0, 8) * 4_8 * cols
       read(u, pos=first) dst(1:cols, 1:rows)
    case (GT_F16)
       allocate(hrow(cols))
       call seek_to(data_pos + dir(idx)%offset + int(row0, 8) * 2_8 * cols)
       do r = 1, rows
          read(u) hrow
          do k = 1, cols
             dst(k, r) = half_bits_to_real(hrow(k))
          end do
       end do
    case (GT_Q4_0)                 ! ggml block_q4_0: f16 d, 16 bytes; lo nibbles = 0..15, hi = 16..31
       allocate(qrow(cols / 32 * 18))
       call seek_to(data_pos + dir(idx)%offset + int(row0, 8) * (cols / 32 * 18))
       do r = 1, rows
          read(u) qrow
          do b = 0, cols / 32 - 1
             d = half_bits_to_real(transfer(qrow(b*18 + 1:b*18 + 2), 0_2))
             do k = 1, 16
                q = iand(int(qrow(b*18 + 2 + k)), 255)
                dst(b*32 + k, r) = real(iand(q, 15) - 8, wp) * d
                dst(b*32 + 16 + k, r) = real(ishft(q, -4) - 8, wp) * d
             end do
          end do
       end do
    case (GT_Q6_K)
       ! ggml block_q6_K, 256 weights in 210 bytes: ql[128] low 4 bits, qh[64] high 2 bits, 16 int8 sub-block scales,
       ! f16 d; weight = d * scale * (q - 32)  (public ggml format, dequantize_row_q6_K; third-party knowledge, not
       ! citable in /root/reference)
       allocate(qrow(cols / 256 * 210))
       call seek_to(data_pos + dir(idx)%offset + int(row0, 8) * (cols / 256 * 210))
       do r = 1, rows
          read(u) qrow
          do b = 0, cols / 256 - 1
             d = half_bits_to_real(transfer(qrow(b*210 + 209:b*210 + 210), 0_2))
             do k = 0, 255
                call q6k_weight(qrow(b*210 + 1:b*210 + 208), k, q, sc)
                dst(b*256 + k + 1, r) = d * real(sc, wp) * real(q, wp)
             end do
          end do
       end do
    case default
       print *, "Type not supported", dir(idx)%ttype
       stop 1
    end select
  end subroutine

  ! position the stream at byte `p` (1-based) without transferring data: the next sequential read starts there
  subroutine seek_to(p)
    integer(8), intent(in) :: p
    read(u, pos=p)
  end subroutine

  ! element k (0..255) of a q6_K block: its 6-bit value minus 32 and its sub-block scale
  pure subroutine q6k_weight(blk, k, q, sc)
    integer(1), intent(in) :: blk(208)       ! ql[128] | qh[64] | scales[16]
    integer, intent(in) :: k
    integer, intent(out) :: q, sc
    integer :: half, l, grp, lo, hi
    half = k / 128                            ! two halves of 128 weights: 64 ql bytes, 32 qh bytes, 8 scales each
    l = mod(k, 32)
    grp = mod(k, 128) / 32                    ! 0..3: which of the four 32-weight quarters
    lo = iand(int(blk(half*64 + mod(grp, 2)*32 + l + 1)), 255)
    if (grp >= 2) lo = ishft(lo, -4)
    lo = iand(lo, 15)
    hi = iand(ishft(iand(int(blk(128 + half*32 + l + 1)), 255), -2*grp), 3)
    q = ior(lo, ishft(hi, 4)) - 32
    sc = int(blk(192 + half*8 + l/16 + 2*grp + 1))      ! int8, signed
  end subroutine

  ! IEEE binary16 bit pattern -> f32 (exact)
  elemental function half_bits_to_real(h) result(x)
    integer(2), intent(in) :: h
    real(kind=wp) :: x
    integer(4) :: bits, s, e, m
    bits = iand(int(h, 4), 65535)
    s = ishft(iand(bits, 32768), 16)
    e = iand(ishft(bits, -10), 31)
    m = iand(bits, 1023)
    if (e == 0) then
       x = real(m, wp) * 2.0_wp**(-24)            ! zero / subnormal
       if (s /= 0) x = -x
    else if (e == 31) then
       x = transfer(ior(s, ior(int(z'7F800000', 4), ishft(m, 13))), x)
    else
       x = transfer(ior(s, ior(ishft(e + 112, 23), ishft(m, 13))), x)
    end if
  end function

end module read_ggml
