! llama2.c-style flat checkpoint ("ak" format, `--ak`).  Same tensor order as the reference's reader
! (/root/reference/llama2.f90:160-292): a 7-int32 header, then f32 arrays
!   token_embedding_table(E,V), rms_att_weight(E,L), wq(E,E,L), wk(E,KV,L), wv(E,KV,L), wo(E,E,L),
!   rms_ffn_weight(E,L), w1(E,H,L), w2(H,E,L), w3(E,H,L), rms_final_weight(E), wcls(E,V)
! into the fused weight_module layout.  Unlike the reference (which ignores the header and uses its
! compile-time dims, llama2.f90:163-164) the header IS the model shape here:
!   emb_dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len
! (a negative vocab_size -- llama2.c's "unshared classifier" flag -- is taken by absolute value).
module ak_loader
  use precision_module
  use weight_module
  implicit none
  private
  public :: load_ak
contains

  subroutine load_ak(filename, w, c, verbose)
    character(len=*), intent(in) :: filename
    type(TransformerWeights), intent(out) :: w
    type(Config), intent(out) :: c
    logical, intent(in) :: verbose
    integer(4) :: hdr(7)
    integer :: u, ios, l, E, H, L_, KV, V

    open(newunit=u, file=filename, form="unformatted", access="stream", status="old", action="read", iostat=ios)
    if (ios /= 0) then
       print *, "cannot open model file ", trim(filename)
       stop 1
    end if
    read(u) hdr
    c%emb_dim = hdr(1); c%hidden_dim = hdr(2); c%n_layers = hdr(3); c%n_heads = hdr(4)
    c%n_kv_heads = hdr(5); c%vocab_size = abs(hdr(6)); c%seq_len = hdr(7)
    if (c%emb_dim <= 0 .or. c%n_heads <= 0 .or. c%n_layers <= 0 .or. c%vocab_size <= 0) then
       print *, "not an ak checkpoint (bad header): ", hdr
       stop 1
    end if
    c%kv_head_size = c%n_kv_heads * (c%emb_dim / c%n_heads)
    E = c%emb_dim; H = c%hidden_dim; L_ = c%n_layers; KV = c%kv_head_size; V = c%vocab_size
    if (verbose) then
       print *, "Embedding dimension: ", E
       print *, "Hidden dimension: ", H
       print *, "Layers: ", L_
       print *, "Heads: ", c%n_heads
       print *, "kv Heads: ", c%n_kv_heads
       print *, "Vocabulary Size: ", V
       print *, "Sequence Length: ", c%seq_len
       print *, "Head Size: ", E / c%n_heads
       print *, "kv Head Size: ", KV
    end if

    w%wtype = 0
    allocate(w%token_embedding_table(E, V), w%rms_att_weight(E, L_), w%wqkv(E, E + 2*KV, L_), w%wo(E, E, L_), &
             w%rms_ffn_weight(E, L_), w%w13(E, 2*H, L_), w%w2(H, E, L_), w%rms_final_weight(E), w%wcls(E, V))
    read(u) w%token_embedding_table
    read(u) w%rms_att_weight
    do l = 1, L_
       read(u) w%wqkv(:, 1:E, l)
    end do
    do l = 1, L_
       read(u) w%wqkv(:, E+1:E+KV, l)
    end do
    do l = 1, L_
       read(u) w%wqkv(:, E+KV+1:E+2*KV, l)
    end do
    read(u) w%wo
    read(u) w%rms_ffn_weight
    do l = 1, L_
       read(u) w%w13(:, 1:H, l)
    end do
    read(u) w%w2
    do l = 1, L_
       read(u) w%w13(:, H+1:2*H, l)
    end do
    read(u) w%rms_final_weight
    read(u, iostat=ios) w%wcls
    if (ios /= 0) w%wcls = w%token_embedding_table      ! llama2.c files with a shared classifier end here
    close(u)
    if (verbose) then
       ! the reference's lines, its counts included: "wv" is ONE layer's slice and "w3" a single hidden row across the layers
       ! (llama2.f90:230 size(wqkv(:,emb+kv+1:,l)), :271 size(w13(:,hidden_dim,:))) -- what -v printed is what -v prints
       print *, "loaded embedding weights:", size(w%token_embedding_table)
       print *, "loaded rms att weights:", size(w%rms_att_weight)
       print *, "loaded wq weights:", E * E * L_
       print *, "loaded wk weights:", E * KV * L_
       print *, "loaded wv weights:", E * KV
       print *, "loaded wo weights:", size(w%wo)
       print *, "loaded rms ffn  weights:", size(w%rms_ffn_weight)
       print *, "loaded w1 weights:", E * H * L_
       print *, "loaded w2 weights:", size(w%w2)
       print *, "loaded w3 weights:", E * L_
       print *, "loaded rms_final weights:", size(w%rms_final_weight)
       if (ios == 0) print *, "loaded wcls weights:", size(w%wcls)      ! (a shared classifier prints nothing, :281-288)
    end if
  end subroutine load_ak

end module ak_loader
