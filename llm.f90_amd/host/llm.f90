! `llm` -- the reference's command line (/root/reference/llama2.f90:4-83, :87-411) on top of the
! MI355X decode path.  Same flags and defaults, same console output; the one call that did all the
! work, `logits = transformer(token,pos,s,weights)` (llama2.f90:380), is now
! `llmk_forward(ctx, token, pos, logits)` into hand-written gfx950 kernels (include/llmk.h).
! Model dims come from the GGUF metadata at run time (the reference hard-codes TinyLlama's,
! llama2.f90:102-108).  There is no CPU forward pass in this program.
!
!   ./llm -m model.gguf [-p prompt] [-n tokens] [-t temperature] [-s tokenizer.bin] [-v]
!         [--ak] [-d device] [--device-argmax] [--prefill] [--timings] [--seed N] [--stream-load] [--ngpu N]
!         [--ngpu N [--tp-rccl]] [--gguf-eps] [--gguf-rope-base] [--encode]
!
! --ngpu N (the 70B configuration, SURVEY.md section 8e): this process becomes rank 0 of N, starts N-1 copies of itself
! (one process per GPU, devices d..d+N-1), every rank loads the file and keeps its shard, the ranks meet through a
! scratch directory (64-byte inbox handles for the one-shot peer-memory collectives, or the RCCL unique id with
! --tp-rccl) and run the same generation loop in lock step; only rank 0 prints.
module arg_parse
  implicit none

  type args
     real :: temperature
     character(:), allocatable :: model_file
     character(:), allocatable :: prompt
     character(:), allocatable :: tokenizer
     logical :: verbose, ak
     logical :: verbose_ext       ! extension: --vx adds this loader's own lines to -v (the matrices' ggml type, streaming)
     integer :: n
     integer :: device            ! extension: HIP device ordinal
     logical :: device_argmax     ! extension: greedy pick on the GPU (SURVEY.md 8f rank 1)
     logical :: prefill           ! extension: the prompt goes through the model as ONE batched pass (llmk_prefill)
     logical :: stream_load       ! extension: matrices go from the file to the device tensor by tensor (always with --ngpu)
     logical :: timings           ! extension: fill the five "Timings" lines from hipEvent section timers (slow path)
     integer :: seed              ! extension: >= 0 seeds the sampler (the reference's is unseeded, llama2.f90:433)
     integer :: ngpu              ! extension: tensor-parallel ranks, one process per GPU
     integer :: tp_rank           ! (internal) rank of a worker started by `--ngpu`
     character(:), allocatable :: tp_dir   ! (internal) rendezvous directory
     logical :: tp_rccl           ! extension: RCCL ring collectives instead of the one-shot peer-memory ones
     logical :: gguf_eps, gguf_rope_base   ! extension: honour the file's rms epsilon / RoPE base (the reference hard-codes
                                           ! 1e-5 and 10000, llama2.f90:454,545)
     logical :: encode_only       ! extension: print the prompt's 1-based token ids (bpe_encode) and stop -- no device needed
  end type args

contains

  subroutine parse_args(a)
    type(args), intent(out) :: a
    integer :: i, nargs
    character(len=1024) :: opt, val

    a%temperature = 0               ! defaults: llama2.f90:26-32
    a%model_file = "stories15M.bin"
    a%prompt = ""
    a%tokenizer = ""
    a%verbose = .false.
    a%verbose_ext = .false.
    a%ak = .false.
    a%n = 256
    a%device = 0
    a%device_argmax = .false.
    a%prefill = .false.
    a%stream_load = .false.
    a%timings = .false.
    a%seed = -1
    a%ngpu = 1
    a%tp_rank = 0
    a%tp_dir = ""
    a%tp_rccl = .false.
    a%gguf_eps = .false.
    a%gguf_rope_base = .false.
    a%encode_only = .false.

    nargs = command_argument_count()
    i = 1
    do while (i <= nargs)
       call get_command_argument(i, opt)
       val = ""
       if (i < nargs) call get_command_argument(i + 1, val)
       select case (trim(opt))
       case ("-m", "--model");       a%model_file = trim(val); i = i + 2
       case ("-p", "--prompt");      a%prompt = trim(val);     i = i + 2
       case ("-s", "--tokenizer");   a%tokenizer = trim(val);  i = i + 2
       case ("-t", "--temperature"); read (val, *) a%temperature; i = i + 2
       case ("-n", "--num_tokens");  read (val, *) a%n;           i = i + 2
       case ("-d", "--device");      read (val, *) a%device;      i = i + 2
       case ("-v", "--verbose");     a%verbose = .true.;       i = i + 1
       case ("--vx");                a%verbose = .true.; a%verbose_ext = .true.; i = i + 1
       case ("--ak");                a%ak = .true.;            i = i + 1
       case ("--device-argmax");     a%device_argmax = .true.; i = i + 1
       case ("--prefill");           a%prefill = .true.;       i = i + 1
       case ("--stream-load");       a%stream_load = .true.;   i = i + 1
       case ("--timings");           a%timings = .true.;       i = i + 1
       case ("--seed");              read (val, *) a%seed;        i = i + 2
       case ("--ngpu");              read (val, *) a%ngpu;        i = i + 2
       case ("--tp-rank");           read (val, *) a%tp_rank;     i = i + 2
       case ("--tp-dir");            a%tp_dir = trim(val);     i = i + 2
       case ("--tp-rccl");           a%tp_rccl = .true.;       i = i + 1
       case ("--gguf-eps");          a%gguf_eps = .true.;      i = i + 1
       case ("--gguf-rope-base");    a%gguf_rope_base = .true.; i = i + 1
       case ("--encode");            a%encode_only = .true.;   i = i + 1
       case default
          print *, "Unrecognized option:", trim(opt)
          stop
       end select
    end do
  end subroutine parse_args

end module arg_parse


! The sink stream_ggml_weights hands its tensors to (read_ggml's tensor_sink): straight to the device context.
module device_sink
  use iso_c_binding
  use llmk_binding
  implicit none
  type(c_ptr) :: sink_ctx = c_null_ptr
contains
  subroutine upload_sink(tensor_id, layer, row_offset, rows, host, nbytes, ggml_type)
    integer, intent(in) :: tensor_id, layer, row_offset, rows, ggml_type
    type(c_ptr), intent(in) :: host
    integer(c_size_t), intent(in) :: nbytes
    call llmk_check(llmk_upload_rows(sink_ctx, int(tensor_id, c_int), int(layer, c_int), int(row_offset, c_int), int(rows, c_int), &
         host, nbytes, int(ggml_type, c_int)), "llmk_upload_rows")
  end subroutine
end module device_sink


! Tokens of llmk_decode_greedy reach the terminal as they are resolved, like the reference's own loop prints them
! (llama2.f90:396): the C-ABI calls ts_on_token from the calling thread, in order.
module token_stream
  use iso_c_binding
  implicit none
  character(:), dimension(:), allocatable :: ts_vocab
  integer(4), allocatable :: ts_len(:)
  logical :: ts_print = .true.
  integer(8) :: ts_first = 0            ! the host's clock_ticks() at the first streamed token (0: none yet)
contains
  subroutine ts_on_token(idx, tok, user) bind(C)
    integer(c_int), value :: idx, tok
    type(c_ptr), value :: user
    if (ts_print) write (*, fmt="(A)", advance="no") ts_vocab(tok)(1:ts_len(tok))
    if (ts_first == 0) call system_clock(ts_first)        ! the same clock as clock_ticks()
  end subroutine
end module token_stream


program llm
  use iso_c_binding
  use token_stream
  use device_sink
  use precision_module
  use weight_module
  use arg_parse
  use read_ggml, only: load_ggml, stream_ggml_weights, verbose_ext
  use ak_loader, only: load_ak
  use llmk_binding
  implicit none

  type(args) :: opts
  type(TransformerWeights), target :: weights
  type(Config) :: conf
  type(RunState) :: s
  type(llmk_config) :: kcfg
  type(c_ptr) :: ctx
  real(kind=wp), allocatable, target :: logits(:)
  real(kind=wp), allocatable :: scores(:), freqs(:), probs(:)
  character(:), dimension(:), allocatable :: vocab
  integer(4), allocatable :: vocab_len(:)
  integer, allocatable :: prompt_tokens(:)
  integer(c_int), allocatable :: batch(:)
  integer(c_int), allocatable :: stream_ids(:)
  integer :: pos0, k, loop_end
  integer :: seq_len, pos, token, next_tok, l, hs, j, max_len
  integer(c_int) :: flags, rc
  integer(8) :: t_start, t_end                      ! clock_ticks(): see elapsed_ms
  real(kind=wp) :: dt_ms
  real(c_float) :: ktimes(5)
  logical :: lead                                   ! this rank prints (rank 0, or the only process)
  integer, allocatable :: hash_tab(:)               ! open-addressing index over vocab (lookup)
  integer :: hash_mask
  real(kind=wp) :: rope_base

  call parse_args(opts)
  lead = opts%tp_rank == 0
  if (opts%ngpu > 1) call tp_launch_workers()
  if (opts%ak) then
     ! llama2.c flat format: no tokenizer inside, `-s tokenizer.bin` is required (llama2.f90:160-356)
     call load_ak(opts%model_file, weights, conf, opts%verbose)
     if (opts%tokenizer == "") then
        print *, "--ak needs a tokenizer file: -s tokenizer.bin"
        stop 1
     end if
  else
     ! --ngpu N (and --stream-load): the matrices stay in the file until the device context exists, then each rank streams
     ! the rows of its own shard to its GPU -- no rank ever holds the model, or even a layer, in host memory
     if (opts%ngpu > 1) opts%stream_load = .true.
     verbose_ext = opts%verbose_ext
     call load_ggml(opts%model_file, weights, conf, vocab, scores, vocab_len, opts%verbose .and. lead, defer=opts%stream_load)
  end if
  if (opts%verbose .and. lead) print *, "Loaded weights"
  if (opts%tokenizer /= "") call read_tokenizer_bin(opts%tokenizer)
  max_len = maxval(vocab_len)
  call build_lookup()
  if (opts%encode_only) then                        ! the tokenizer alone (llama2.f90:372): ids on one line, no device
     prompt_tokens = bpe_encode(opts%prompt)
     print '(*(I0,1X))', prompt_tokens
     stop
  end if
  if (opts%seed >= 0) call seed_sampler(opts%seed)

  ! ---- device context + one-time weight upload (the host arrays are not needed afterwards) -----
  ! The reference's five section timers (llama2.f90:538-638) need an event pair and a sync per section, which rules out
  ! the persistent token kernel and the hipGraph: they are opt-in (--timings) so that `-v`, like the reference's, only
  ! adds prints and the tokens/second line always describes the fast path.  Without --timings the five lines print 0.
  flags = 0
  if (opts%timings) flags = LLMK_FLAG_TIMINGS
  kcfg = llmk_config(conf%emb_dim, conf%hidden_dim, conf%n_layers, conf%n_heads, conf%n_kv_heads, &
                     conf%vocab_size, conf%seq_len, weights%wtype, tp_device(), flags)
  if (opts%ngpu > 1) then
     call llmk_check(llmk_create_tp(kcfg, int(opts%tp_rank, c_int), int(opts%ngpu, c_int), ctx), "llmk_create_tp")
  else
     call llmk_check(llmk_create(kcfg, ctx), "llmk_create")
  end if
  if (weights%wcls_type /= weights%wtype) &
       call llmk_check(llmk_set_tensor_type(ctx, LLMK_WCLS, int(weights%wcls_type, c_int)), "llmk_set_tensor_type")
  if (opts%stream_load .and. .not. opts%ak) then
     sink_ctx = ctx
     call stream_ggml_weights(upload_sink, weights, conf, opts%tp_rank, max(opts%ngpu, 1), opts%verbose .and. lead)
  else
     call upload_weights()
  end if
  if (opts%ngpu > 1) call tp_connect()
  if (opts%gguf_eps .and. conf%rms_eps > 0) call llmk_check(llmk_set_rms_eps(ctx, conf%rms_eps), "llmk_set_rms_eps")

  ! RoPE frequencies with the reference's own expression (llama2.f90:544-545): for 1-based odd i,
  ! head_dim = mod(i,head_size) = 1,3,5,...; freq = 1/10000**(head_dim/head_size)
  hs = conf%emb_dim / conf%n_heads
  allocate(freqs(hs / 2))
  rope_base = 10000.0
  if (opts%gguf_rope_base .and. conf%rope_freq_base > 0) rope_base = conf%rope_freq_base
  do j = 1, hs / 2
     freqs(j) = 1.0 / (rope_base ** (real(2*j - 1, kind=wp) / hs))
  end do
  call llmk_check(llmk_set_rope_freqs(ctx, freqs, int(hs / 2, c_int)), "llmk_set_rope_freqs")
  ! which of the two implementations of the pass this shape and weight type get (include/llmk.h llmk_path): the persistent
  ! whole-token kernel exists for the compiled-in shapes only (the reference's dims are compile-time too, llama2.f90:102-108);
  ! every other model runs -- correctly, at roughly two thirds of the rate -- on five launches per layer.  Said under --vx.
  if (opts%verbose_ext .and. lead) then
     select case (llmk_path(ctx))
     case (1); print *, "device path: persistent whole-token kernel"
     case (2); print *, "device path: tensor-parallel, one-shot peer-memory collectives"
     case (3); print *, "device path: tensor-parallel, RCCL collectives"
     case default; print *, "device path: five kernels per layer (no persistent kernel is instantiated for this shape and weight type)"
     end select
     ! ... and for which shapes this build of libllmk.so has one (make TK_SHAPES="E,H,NH,NKV,V,WT ..." adds more; the reference's
     ! own dims are seven compile-time parameters, llama2.f90:102-108)
     block
       character(kind=c_char) :: sbuf(4096)
       character(len=4096) :: stxt
       integer :: k
       if (llmk_tk_shapes(sbuf, int(size(sbuf), c_size_t)) == 0) then
          stxt = ""
          do k = 1, size(sbuf)
             if (sbuf(k) == c_null_char) exit
             stxt(k:k) = sbuf(k)
          end do
          print *, "persistent kernel built for (E,H,heads,kv heads,V,type): ", trim(stxt)
       end if
     end block
  end if

  allocate(logits(conf%vocab_size), probs(conf%vocab_size))
  s%times = 0

  seq_len = conf%seq_len
  if (opts%n <= seq_len) then                        ! llama2.f90:363-368
     seq_len = opts%n
  else if (lead) then
     print *, opts%n, "greater than maxinum squence length"
     print *, "set to", seq_len
  end if

  prompt_tokens = bpe_encode(opts%prompt)

  ! ---- generation loop (llama2.f90:376-402) -------------------------------------------------------
  t_start = 0
  token = 2                                          ! BOS: 1-based index of <s>
  pos0 = 1
  if (opts%prefill .and. size(prompt_tokens) > 0 .and. size(prompt_tokens) < seq_len) then
     ! positions 1 .. k+1 of the loop below (BOS, then the k prompt tokens) in one call: same cache, same logits
     ! at position k+1, same text on stdout
     k = size(prompt_tokens)
     allocate(batch(k + 1))
     batch(1) = 2
     batch(2:) = int(prompt_tokens, c_int)
     t_start = clock_ticks()   ! the clock covers the prompt pass: tokens/second counts those positions too (llama2.f90:405)
     call llmk_check(llmk_prefill(ctx, batch, int(k + 1, c_int), 1_c_int, logits), "llmk_prefill")
     if (lead) then
        do pos = 1, k
           write (*, fmt="(A)", advance="no") vocab(prompt_tokens(pos))(1:vocab_len(prompt_tokens(pos)))
        end do
     end if
     if (opts%temperature == 0) then
        token = argmax1(logits)
     else
        probs = softmax_t(logits / opts%temperature)
        token = sample(probs)
     end if
     if (lead) write (*, fmt="(A)", advance="no") vocab(token)(1:vocab_len(token))
     pos0 = k + 2
  end if
  ! --device-argmax at temperature 0: every position after the prompt is one call (llmk_decode_greedy: the argmax stays on
  ! the device, the launches are enqueued back to back, the ids stream back through mapped memory and are printed as they
  ! arrive).  The positions whose next token is a PROMPT token still go through the loop below -- and so does the FIRST
  ! position in any case: the reference starts its clock after the first token (llama2.f90:398-399) and divides seq_len - 1
  ! tokens by what follows, so that token is produced, and the clock started, before the pipelined launches are enqueued
  ! (started at the first streamed id instead, several tokens had already completed: the printed rate was slightly high).
  loop_end = seq_len
  if (opts%device_argmax .and. opts%temperature == 0 .and. opts%ngpu == 1) loop_end = min(seq_len, max(size(prompt_tokens), pos0))
  do pos = pos0, loop_end
     call llmk_check(llmk_forward(ctx, int(token, c_int), int(pos, c_int), logits), "llmk_forward")
     if (pos <= size(prompt_tokens)) then
        next_tok = prompt_tokens(pos)
     else if (opts%temperature == 0) then
        next_tok = argmax1(logits)
     else
        probs = softmax_t(logits / opts%temperature)
        next_tok = sample(probs)
     end if
     token = next_tok
     if (lead) write (*, fmt="(A)", advance="no") vocab(token)(1:vocab_len(token))
     if (t_start == 0) t_start = clock_ticks()       ! clock starts after the first token
  end do
  if (loop_end < seq_len) then
     allocate(stream_ids(seq_len - loop_end))
     ts_vocab = vocab
     ts_len = vocab_len
     ts_print = lead
     call llmk_check(llmk_decode_greedy(ctx, int(token, c_int), int(loop_end + 1, c_int), int(seq_len - loop_end, c_int), &
          stream_ids, c_funloc(ts_on_token), c_null_ptr), "llmk_decode_greedy")
     token = stream_ids(size(stream_ids))
     if (t_start == 0) t_start = ts_first
  end if
  t_end = clock_ticks()
  dt_ms = elapsed_ms(t_start, t_end)

  call llmk_check(llmk_timings(ctx, ktimes), "llmk_timings")
  s%times = ktimes
  if (lead) then
     print *, ""
     print *, "Inference time: ", dt_ms / 1000, " seconds"
     print *, 1000 * (seq_len - 1) / dt_ms, "tokens/second"
     print *, "Timings"
     do l = 1, 5
        print *, l, s%times(l) / seq_len
     end do
  end if
  if (opts%ngpu > 1) call tp_finish()
  rc = llmk_destroy(ctx)

contains

  subroutine upload_weights()
    integer(c_size_t) :: f4
    f4 = 4
    call llmk_check(llmk_upload(ctx, LLMK_TOKEN_EMBEDDING_TABLE, c_loc(weights%token_embedding_table), &
         f4 * size(weights%token_embedding_table, kind=c_size_t), LLMK_TYPE_F32), "upload token_embedding_table")
    call llmk_check(llmk_upload(ctx, LLMK_RMS_ATT_WEIGHT, c_loc(weights%rms_att_weight), &
         f4 * size(weights%rms_att_weight, kind=c_size_t), LLMK_TYPE_F32), "upload rms_att_weight")
    call llmk_check(llmk_upload(ctx, LLMK_RMS_FFN_WEIGHT, c_loc(weights%rms_ffn_weight), &
         f4 * size(weights%rms_ffn_weight, kind=c_size_t), LLMK_TYPE_F32), "upload rms_ffn_weight")
    call llmk_check(llmk_upload(ctx, LLMK_RMS_FINAL_WEIGHT, c_loc(weights%rms_final_weight), &
         f4 * size(weights%rms_final_weight, kind=c_size_t), LLMK_TYPE_F32), "upload rms_final_weight")
    if (weights%wtype == LLMK_TYPE_F32) then
       call llmk_check(llmk_upload(ctx, LLMK_WQKV, c_loc(weights%wqkv), f4 * size(weights%wqkv, kind=c_size_t), &
            LLMK_TYPE_F32), "upload wqkv")
       call llmk_check(llmk_upload(ctx, LLMK_WO, c_loc(weights%wo), f4 * size(weights%wo, kind=c_size_t), &
            LLMK_TYPE_F32), "upload wo")
       call llmk_check(llmk_upload(ctx, LLMK_W13, c_loc(weights%w13), f4 * size(weights%w13, kind=c_size_t), &
            LLMK_TYPE_F32), "upload w13")
       call llmk_check(llmk_upload(ctx, LLMK_W2, c_loc(weights%w2), f4 * size(weights%w2, kind=c_size_t), &
            LLMK_TYPE_F32), "upload w2")
       call llmk_check(llmk_upload(ctx, LLMK_WCLS, c_loc(weights%wcls), f4 * size(weights%wcls, kind=c_size_t), &
            LLMK_TYPE_F32), "upload wcls")
       deallocate(weights%wqkv, weights%wo, weights%w13, weights%w2, weights%wcls)
    else
       call llmk_check(llmk_upload(ctx, LLMK_WQKV, c_loc(weights%wqkv_raw), size(weights%wqkv_raw, kind=c_size_t), &
            int(weights%wtype, c_int)), "upload wqkv")
       call llmk_check(llmk_upload(ctx, LLMK_WO, c_loc(weights%wo_raw), size(weights%wo_raw, kind=c_size_t), &
            int(weights%wtype, c_int)), "upload wo")
       call llmk_check(llmk_upload(ctx, LLMK_W13, c_loc(weights%w13_raw), size(weights%w13_raw, kind=c_size_t), &
            int(weights%wtype, c_int)), "upload w13")
       call llmk_check(llmk_upload(ctx, LLMK_W2, c_loc(weights%w2_raw), size(weights%w2_raw, kind=c_size_t), &
            int(weights%wtype, c_int)), "upload w2")
       if (weights%wcls_type == LLMK_TYPE_F32) then      ! classifier dequantised by the loader (q6_K output.weight)
          call llmk_check(llmk_upload(ctx, LLMK_WCLS, c_loc(weights%wcls), f4 * size(weights%wcls, kind=c_size_t), &
               LLMK_TYPE_F32), "upload wcls")
          deallocate(weights%wcls)
       else
          call llmk_check(llmk_upload(ctx, LLMK_WCLS, c_loc(weights%wcls_raw), size(weights%wcls_raw, kind=c_size_t), &
               int(weights%wcls_type, c_int)), "upload wcls")
          deallocate(weights%wcls_raw)
       end if
       deallocate(weights%wqkv_raw, weights%wo_raw, weights%w13_raw, weights%w2_raw)
    end if
    deallocate(weights%token_embedding_table)
  end subroutine upload_weights

  ! maxloc(v, dim=1) (llama2.f90:388: the first maximum, 1-based) in two vectorisable passes: eight running maxima, then the first
  ! element equal to the largest.  flang's maxloc is a scalar compare-and-branch loop: 20-45 us for the 32,000 logits, 3-6 % of a
  ! 0.68 ms token -- the one piece of host arithmetic on the path (measured: bench.py fortran_host, profiles/r05_*fortran_cli*).
  function argmax1(v) result(idx)
    real(kind=wp), intent(in) :: v(:)
    integer :: idx, i, n8
    real(kind=wp) :: m(8), mm
    n8 = size(v) / 8 * 8
    m = v(1)
    do i = 1, n8, 8
       m = max(m, v(i:i + 7))
    end do
    mm = maxval(m)
    do i = n8 + 1, size(v)
       mm = max(mm, v(i))
    end do
    do idx = 1, size(v)
       if (v(idx) == mm) return
    end do
    idx = maxloc(v, dim=1)                    ! (a NaN among the logits: whatever the intrinsic answers)
  end function argmax1

  ! Wall clock.  The reference converts a 4-byte millisecond count to real(4) BEFORE subtracting (llama2.f90:417-423):
  ! a count of ~1e9 has a 64-128 ms quantum as a float, which is nothing against its 60 s runs and everything against a
  ! 0.17 s run of 256 tokens on the GPU.  Same two printed lines, same real(4) list-directed format; the difference is taken
  ! in 8-byte ticks first (microseconds or finer with amdflang).
  function clock_ticks() result(t)
    integer(8) :: t
    call system_clock(t)
    if (t == 0) t = 1                    ! 0 means "clock not started yet" in the generation loop
  end function clock_ticks
  function elapsed_ms(t0, t1) result(ms)
    integer(8), intent(in) :: t0, t1
    real(kind=wp) :: ms
    integer(8) :: rate
    call system_clock(count_rate=rate)
    ms = real(1000.0d0 * real(t1 - t0, 8) / real(rate, 8), kind=wp)
  end function elapsed_ms

  ! llama2.c tokenizer.bin: max_len, then (f32 score, i32 len, bytes) per token (llama2.f90:321-356)
  subroutine read_tokenizer_bin(path)
    character(len=*), intent(in) :: path
    integer :: tu, n, tl, width
    real(kind=wp) :: sc
    character(len=:), allocatable :: buf
    open(newunit=tu, file=path, form="unformatted", access="stream", status="old", action="read")
    read(tu) width
    if (allocated(vocab)) deallocate(vocab, scores, vocab_len)
    allocate(character(len=width) :: vocab(conf%vocab_size))
    allocate(scores(conf%vocab_size), vocab_len(conf%vocab_size))
    do n = 1, conf%vocab_size
       read(tu) sc
       read(tu) tl
       allocate(character(len=tl) :: buf)
       read(tu) buf
       vocab(n) = buf
       scores(n) = sc
       vocab_len(n) = tl
       deallocate(buf)
    end do
    close(tu)
  end subroutine read_tokenizer_bin

  ! ---- vocabulary index: exact-match lookup honouring the true token length (trailing blanks are data).  The reference
  ! scans all V strings per lookup (llama2.f90:643-655), O(V) per candidate pair of every merge round; this is an
  ! open-addressing hash (FNV-1a over the bytes) built once.  Same answer: the FIRST index holding the string.
  function hash_bytes(str, n) result(h)
    character(len=*), intent(in) :: str
    integer, intent(in) :: n
    integer :: h, i
    integer(8) :: acc
    acc = 2166136261_8
    do i = 1, n
       acc = iand(ieor(acc, int(ichar(str(i:i)), 8)) * 16777619_8, 4294967295_8)
    end do
    h = int(iand(acc, int(hash_mask, 8)))
  end function hash_bytes

  subroutine build_lookup()
    integer :: i, h, cap
    cap = 1
    do while (cap < 2 * size(vocab) + 2)
       cap = cap * 2
    end do
    hash_mask = cap - 1
    if (allocated(hash_tab)) deallocate(hash_tab)
    allocate(hash_tab(0:cap - 1))
    hash_tab = 0
    do i = 1, size(vocab)
       h = hash_bytes(vocab(i), int(vocab_len(i)))
       do
          if (hash_tab(h) == 0) then
             hash_tab(h) = i
             exit
          end if
          if (vocab_len(hash_tab(h)) == vocab_len(i)) then
             if (vocab(hash_tab(h))(1:vocab_len(i)) == vocab(i)(1:vocab_len(i))) exit    ! duplicate: the first index wins
          end if
          h = iand(h + 1, hash_mask)
       end do
    end do
  end subroutine build_lookup

  function lookup(str, n) result(idx)
    character(len=*), intent(in) :: str
    integer, intent(in) :: n
    integer :: idx, h
    h = hash_bytes(str, n)
    do
       idx = hash_tab(h)
       if (idx == 0) exit
       if (vocab_len(idx) == n) then
          if (vocab(idx)(1:n) == str(1:n)) return
       end if
       h = iand(h + 1, hash_mask)
    end do
    idx = -1
  end function lookup

  ! a byte with no single-character token: llama's byte-fallback token "<0xXX>", else <unk> (the reference indexes
  ! vocab_len(-1) here, llama2.f90:666-668 + :683)
  function byte_token(ch) result(idx)
    character(len=1), intent(in) :: ch
    integer :: idx
    character(len=6) :: name
    character(len=16), parameter :: hex = "0123456789ABCDEF"
    integer :: b
    idx = lookup(ch, 1)
    if (idx > 0) return
    b = ichar(ch)
    name = "<0x" // hex(b / 16 + 1:b / 16 + 1) // hex(mod(b, 16) + 1:mod(b, 16) + 1) // ">"
    idx = lookup(name, 6)
    if (idx < 0) idx = 1                                   ! <unk>
  end function byte_token

  ! llama2.c-style BPE (behaviour of llama2.f90:658-724): start from one token per byte, then
  ! repeatedly fuse the adjacent pair whose concatenation is the best-scoring vocabulary entry.
  function bpe_encode(text) result(tokens)
    character(len=*), intent(in) :: text
    integer, allocatable :: tokens(:)
    integer :: n, i, best_i, best_tok, cand, la, lb
    real(kind=wp) :: best_score
    character(len=:), allocatable :: pair

    n = len(text)
    allocate(tokens(n))
    do i = 1, n
       tokens(i) = byte_token(text(i:i))
    end do
    do
       best_score = -1e10
       best_i = -1
       best_tok = -1
       do i = 1, n - 1
          la = vocab_len(tokens(i))
          lb = vocab_len(tokens(i + 1))
          pair = vocab(tokens(i))(1:la) // vocab(tokens(i + 1))(1:lb)
          cand = lookup(pair, la + lb)
          if (cand > 0) then
             if (scores(cand) > best_score) then
                best_score = scores(cand)
                best_i = i
                best_tok = cand
             end if
          end if
       end do
       if (best_i < 0) exit
       tokens(best_i) = best_tok
       tokens(best_i + 1:n - 1) = tokens(best_i + 2:n)
       n = n - 1
    end do
    tokens = tokens(1:n)
  end function bpe_encode

  ! softmax over the whole vocabulary for temperature sampling (llama2.f90:390, :468-478)
  function softmax_t(x) result(p)
    real(kind=wp), intent(in) :: x(:)
    real(kind=wp) :: p(size(x))
    p = exp(x - maxval(x))
    p = p / sum(p)
  end function softmax_t

  ! inverse-CDF draw (llama2.f90:428-447)
  function sample(p) result(idx)
    real(kind=wp), intent(in) :: p(:)
    integer :: idx
    real(kind=wp) :: r, cdf
    call random_number(r)
    cdf = 0
    do idx = 1, size(p)
       cdf = cdf + p(idx)
       if (r < cdf) return
    end do
    idx = size(p)
  end function sample

  ! `--seed N`: a reproducible sampler (and identical draws on every tensor-parallel rank)
  subroutine seed_sampler(seed)
    integer, intent(in) :: seed
    integer :: n, i
    integer, allocatable :: put(:)
    call random_seed(size=n)
    allocate(put(n))
    do i = 1, n
       put(i) = ieor(seed * 1664525 + 1013904223, i * 668265263)
    end do
    call random_seed(put=put)
  end subroutine seed_sampler

  ! ---- `--ngpu N`: one process per GPU ------------------------------------------------------------------------------
  integer(c_int) function tp_device()
    character(len=8) :: env
    integer :: st
    tp_device = int(opts%device + opts%tp_rank, c_int)
    call get_environment_variable("LLMK_TP_SAME_DEVICE", env, status=st)      ! test aid: all ranks on one GPU
    if (st == 0) tp_device = int(opts%device, c_int)
  end function tp_device

  function shell_quote(a) result(q)
    character(len=*), intent(in) :: a
    character(len=:), allocatable :: q
    integer :: i
    q = "'"
    do i = 1, len(a)
       if (a(i:i) == "'") then
          q = q // "'\''"
       else
          q = q // a(i:i)
       end if
    end do
    q = q // "'"
  end function shell_quote

  ! rank 0: make the rendezvous directory and start ranks 1..N-1 as copies of this command line
  subroutine tp_launch_workers()
    character(kind=c_char) :: tmpl(32)
    character(len=4096) :: arg
    character(len=:), allocatable :: cmd, base
    character(len=32) :: num
    integer :: i, r, seed, ticks
    if (opts%tp_rank > 0) return                       ! a worker: everything was handed down
    ! a fresh private directory (mkdtemp: unpredictable name, mode 0700, fails rather than reuse): nothing stale from a
    ! crashed run with the same pid, nothing another local user could have planted
    tmpl(1:22) = transfer("/tmp/llmk_tp_XXXXXX" // c_null_char // "  ", tmpl(1:22))
    if (.not. c_associated(c_mkdtemp(tmpl))) then
       print *, "cannot create the tensor-parallel rendezvous directory under /tmp"
       stop 1
    end if
    opts%tp_dir = transfer(tmpl(1:19), repeat(" ", 19))
    call execute_command_line("touch " // opts%tp_dir // "/alive")
    call get_command_argument(0, arg)
    base = shell_quote(trim(arg))
    do i = 1, command_argument_count()
       call get_command_argument(i, arg)
       base = base // " " // shell_quote(trim(arg))
    end do
    if (opts%temperature /= 0 .and. opts%seed < 0) then   ! sampling: every rank must draw the same numbers
       call system_clock(ticks)
       seed = iand(ticks, 1073741823)
       opts%seed = seed
       write (num, "(I0)") seed
       base = base // " --seed " // trim(num)
    end if
    do r = 1, opts%ngpu - 1
       write (num, "(I0)") r
       cmd = base // " --tp-rank " // trim(num) // " --tp-dir " // opts%tp_dir // " > /dev/null 2> " // opts%tp_dir // &
             "/rank" // trim(num) // ".err &"
       call execute_command_line(cmd, wait=.false.)
    end do
  end subroutine tp_launch_workers

  subroutine tp_put(name, bytes)
    character(len=*), intent(in) :: name
    character(kind=c_char), intent(in) :: bytes(:)
    integer :: fu
    open(newunit=fu, file=opts%tp_dir // "/" // name // ".tmp", form="unformatted", access="stream", status="replace")
    write (fu) bytes
    close(fu)
    call execute_command_line("mv " // opts%tp_dir // "/" // name // ".tmp " // opts%tp_dir // "/" // name)   ! atomic
  end subroutine tp_put

  subroutine tp_get(name, bytes)
    character(len=*), intent(in) :: name
    character(kind=c_char), intent(out) :: bytes(:)
    integer :: fu, waited
    logical :: there
    waited = 0
    do
       inquire(file=opts%tp_dir // "/" // name, exist=there)
       if (there) exit
       inquire(file=opts%tp_dir // "/alive", exist=there)
       if (.not. there .and. opts%tp_rank > 0) stop 1   ! rank 0 is gone (it removes the directory on every exit path it controls)
       rc = c_usleep(20000_c_int)
       waited = waited + 1
       if (waited > 15000) then                        ! 5 minutes: a peer died while loading
          print *, "tensor-parallel rendezvous timed out waiting for ", name
          if (opts%tp_rank == 0) then                  ! what the workers said, then nothing left behind
             call execute_command_line("cat " // opts%tp_dir // "/rank*.err 1>&2; rm -rf " // opts%tp_dir)
          end if
          stop 1
       end if
    end do
    open(newunit=fu, file=opts%tp_dir // "/" // name, form="unformatted", access="stream", status="old", action="read")
    read (fu) bytes
    close(fu)
  end subroutine tp_get

  ! exchange the inbox handles (or the RCCL id) through the rendezvous directory
  subroutine tp_connect()
    character(kind=c_char) :: h(64), uid(128), verdict(1)
    character(kind=c_char), allocatable :: all(:)
    character(len=32) :: num
    integer :: r
    logical :: all_ok
    if (opts%tp_rccl) then
       if (opts%tp_rank == 0) then
          call llmk_check(llmk_tp_unique_id(uid), "llmk_tp_unique_id")
          call tp_put("uid", uid)
       else
          call tp_get("uid", uid)
       end if
       call llmk_check(llmk_tp_init_comm(ctx, uid), "llmk_tp_init_comm")
       return
    end if
    call llmk_check(llmk_tp_p2p_handle(ctx, h), "llmk_tp_p2p_handle")
    write (num, "(I0)") opts%tp_rank
    call tp_put("inbox" // trim(num), h)
    allocate(all(64 * opts%ngpu))
    do r = 0, opts%ngpu - 1
       write (num, "(I0)") r
       call tp_get("inbox" // trim(num), all(64 * r + 1:64 * r + 64))
    end do
    call llmk_check(llmk_tp_p2p_connect(ctx, all), "llmk_tp_p2p_connect")
    ! The peer-memory path proves itself on THIS hardware before the first token (64 rounds of every collective on known
    ! integers).  Unless every rank passes, all ranks drop it together and the token pass runs over RCCL.
    verdict(1) = "y"
    if (llmk_tp_p2p_selftest(ctx, 64_c_int) /= 0) verdict(1) = "n"
    write (num, "(I0)") opts%tp_rank
    call tp_put("p2p" // trim(num), verdict)
    all_ok = .true.
    do r = 0, opts%ngpu - 1
       write (num, "(I0)") r
       call tp_get("p2p" // trim(num), verdict)
       if (verdict(1) /= "y") all_ok = .false.
    end do
    if (.not. all_ok) then
       if (lead) write (0, *) "llmk: the peer-memory collectives failed their self-test on this node: using RCCL"
       call llmk_check(llmk_tp_p2p_disable(ctx), "llmk_tp_p2p_disable")
       if (opts%tp_rank == 0) then
          call llmk_check(llmk_tp_unique_id(uid), "llmk_tp_unique_id")
          call tp_put("uid", uid)
       else
          call tp_get("uid", uid)
       end if
       call llmk_check(llmk_tp_init_comm(ctx, uid), "llmk_tp_init_comm")
    end if
  end subroutine tp_connect

  ! rank 0 leaves last and removes the directory
  subroutine tp_finish()
    character(kind=c_char) :: one(1)
    character(len=32) :: num
    integer :: r
    one(1) = "x"
    write (num, "(I0)") opts%tp_rank
    if (opts%tp_rank > 0) then
       call tp_put("done" // trim(num), one)
       return
    end if
    do r = 1, opts%ngpu - 1
       write (num, "(I0)") r
       call tp_get("done" // trim(num), one)
    end do
    call execute_command_line("rm -rf " // opts%tp_dir)
  end subroutine tp_finish

end program llm
