! Test utility (tests/test_host_cpu.py): load a GGUF with the product's Fortran loader
! (llm.f90_amd/host/gguf_loader.f90) and dump config, tokenizer and every fused weight array to a
! flat binary file that the python test compares against its own reading of the same file.
program loader_dump
  use precision_module
  use weight_module
  use read_ggml, only: load_ggml
  implicit none
  type(TransformerWeights) :: w
  type(Config) :: c
  character(:), dimension(:), allocatable :: vocab
  real(kind=wp), allocatable :: scores(:)
  integer(4), allocatable :: tl(:)
  character(len=1024) :: inpath, outpath
  integer :: u, i

  call get_command_argument(1, inpath)
  call get_command_argument(2, outpath)
  call load_ggml(trim(inpath), w, c, vocab, scores, tl, .false.)
  open(newunit=u, file=trim(outpath), form="unformatted", access="stream", status="replace")
  write(u) c%emb_dim, c%hidden_dim, c%n_layers, c%n_heads, c%n_kv_heads, c%vocab_size, c%seq_len, c%kv_head_size
  write(u) w%wtype, int(len(vocab(1)), 4), w%wcls_type
  write(u) c%rms_eps, c%rope_freq_base
  write(u) tl
  write(u) scores
  do i = 1, size(vocab)
     write(u) vocab(i)
  end do
  write(u) w%token_embedding_table
  write(u) w%rms_att_weight
  write(u) w%rms_ffn_weight
  write(u) w%rms_final_weight
  if (w%wtype == 0) then
     write(u) w%wqkv
     write(u) w%wo
     write(u) w%w13
     write(u) w%w2
     write(u) w%wcls
  else
     write(u) w%wqkv_raw
     write(u) w%wo_raw
     write(u) w%w13_raw
     write(u) w%w2_raw
     if (w%wcls_type == 0) then
        write(u) w%wcls            ! dequantised by the loader (q6_K output.weight)
     else
        write(u) w%wcls_raw
     end if
  end if
  close(u)
end program loader_dump
