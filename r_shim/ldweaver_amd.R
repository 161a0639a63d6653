# Drop-in R definitions on top of the shim (same signatures as the reference).  NOT run in this repository
# (no R in the build image); the Python mirror ldweaver_amd/mi.py implements and tests the same logic.

.ldwamd_states_from_snpdat <- function(snp.dat) {
  # five L x N one-hot sparse matrices -> one raw L x N matrix, row-major, values 0..4
  st <- matrix(as.raw(0), nrow = snp.dat$nseq, ncol = snp.dat$nsnp)           # N x L column-major == L x N row-major
  for (k in 1:4) {
    m <- Matrix::t(snp.dat[[c("snp.matrix_C", "snp.matrix_G", "snp.matrix_T", "snp.matrix_N")[k]]])  # N x L
    idx <- Matrix::which(m)
    st[idx] <- as.raw(k)
  }
  st
}

# r05: options(ldwamd.devices = 0:7) -> one context per GPU of the node; the block loop (R/computePairwiseMI.R:103-116) and the
# sequence x sequence comparison of the Hamming weights are then shared over them INSIDE the library (ldw_mi_all_pairs_multi,
# ldw_hamming_weights_multi: worker threads, peer-to-peer gather into device ldwamd.devices[1]); results do not depend on it.
# options(ldwamd.sr_rows_stay = TRUE) (with several devices): the short-range rows — 99 % of the links — are NOT gathered; every device keeps
# the rows it computed, perform_MI_computation takes mergeNsort_sr_links_device below and the library runs the model's reductions over the
# devices (ldw_sr_len_quantiles_multi / _excess_stats_multi / _pvalues_multi); same sr_links.tsv.
.ldwamd_devices_set <- NULL
.ldwamd_use_devices <- function() {
  devs <- as.integer(getOption("ldwamd.devices", 0L))
  if (!identical(devs, .ldwamd_devices_set)) {
    .Call("ldwamd_set_devices", devs)
    .ldwamd_devices_set <<- devs
  }
  .Call("ldwamd_set_sr_rows_stay", isTRUE(getOption("ldwamd.sr_rows_stay", FALSE)) && length(devs) > 1)
  invisible(devs)
}

# r06: between jobs of a long session (or before another package allocates on the same GPUs): destroy the contexts and give the library's pooled device
# blocks and host buffers back; the next call makes its contexts again.  Returns the bytes released.
ldwamd_release <- function() {
  n <- .Call("ldwamd_release")
  .ldwamd_devices_set <<- NULL
  invisible(n)
}

.ACGTN2num <- function(nv, cv, ncores) invisible(.Call("ldwamd_ACGTN2num", nv, cv, as.integer(ncores)))

estimate_Hamming_distance_weights <- function(snp.dat, threshold = 0.1, mega_dset = F) {
  t0 <- Sys.time()
  .ldwamd_use_devices()
  .Call("ldwamd_set_alignment", .ldwamd_states_from_snpdat(snp.dat), snp.dat$nsnp, snp.dat$nseq)
  hdw <- .Call("ldwamd_hamming_weights", as.integer(snp.dat$nsnp * threshold), snp.dat$nseq)
  names(hdw) <- snp.dat$seq.names
  cat(paste("Done in", round(difftime(Sys.time(), t0, units = "secs"), 2), "s\n"))
  hdw
}

perform_MI_computation <- function(snp.dat, hdw, cds_var, ncores, lr_save_path = NULL, sr_save_path = NULL, plt_folder = NULL,
                                   sr_dist = 20000, lr_retain_links = 1e6, max_blk_sz = 10000, srp_cutoff = 3, runARACNE = TRUE,
                                   perform_SR_analysis_only = FALSE, order_links = T, mega_dset = F) {
  if (is.null(lr_save_path)) lr_save_path <- file.path(getwd(), "lr_links.tsv")
  if (is.null(sr_save_path)) sr_save_path <- file.path(getwd(), "sr_links.tsv")
  if (is.null(plt_folder)) plt_folder <- file.path(getwd(), "PLOTS")
  if (!file.exists(plt_folder)) dir.create(plt_folder)
  max_blk_sz <- round(max_blk_sz, -3)
  MI_cmp_blks <- make_blocks(snp.dat$nsnp, max_blk_sz)                       # reference's own helper, unchanged
  lr_links_approx <- 1
  if (!perform_SR_analysis_only) {                                           # R/computePairwiseMI.R:94-97, stays in R
    snp_subset <- min(snp.dat$nsnp, round(snp.dat$nsnp * 0.1))
    set.seed(1988)
    cnt <- sapply(snp.dat$POS[sample(snp.dat$nsnp, snp_subset)],
                  function(x) sum((0.5 * snp.dat$g - abs((x - snp.dat$POS) %% snp.dat$g - 0.5 * snp.dat$g)) > sr_dist))
    lr_links_approx <- sum(cnt) / snp_subset * snp.dat$nsnp / 2
  }
  .ldwamd_use_devices()
  .Call("ldwamd_set_alignment", .ldwamd_states_from_snpdat(snp.dat), snp.dat$nsnp, snp.dat$nseq)
  .Call("ldwamd_ctx_reserve", snp.dat$nsnp, snp.dat$nseq, max_blk_sz)        # r04: the pass's buffers, on a side thread, while the weights are set
  .Call("ldwamd_set_weights", as.numeric(hdw))
  .Call("ldwamd_set_snp_meta", as.numeric(snp.dat$r), as.raw(t(snp.dat$uqe)), as.integer(snp.dat$POS),
        as.integer(cds_var$paint), as.numeric(snp.dat$g))
  # r05: with options(ldwamd.native_tsv = TRUE) on ONE device lr_links.tsv is appended while the block loop runs, item by item — the reference's
  # per-block write.table(append = T) (:362): a job that dies leaves the finished blocks' rows
  stream_lr <- isTRUE(getOption("ldwamd.native_tsv", FALSE)) && !perform_SR_analysis_only && length(getOption("ldwamd.devices", 0L)) == 1
  if (stream_lr) .Call("ldwamd_lr_stream_begin", lr_save_path)
  res <- tryCatch(.Call("ldwamd_mi_all_pairs", as.integer(t(as.matrix(MI_cmp_blks))), sr_dist, lr_retain_links, lr_links_approx,
                        perform_SR_analysis_only, 0L),
                  error = function(e) { if (stream_lr) try(.Call("ldwamd_lr_stream_end"), silent = TRUE); stop(e) })
  if (stream_lr) .Call("ldwamd_lr_stream_end")
  to_df <- function(t) {
    # POS keeps its type (an INTEGER vector in the reference, src/getACGTNsites.cpp:97,173: pos1 / pos2 print as integers)
    pos2 <- snp.dat$POS[t[[1]] + 1]; pos1 <- snp.dat$POS[t[[2]] + 1]
    data.frame(pos1 = pos1, pos2 = pos2, clust1 = cds_var$paint[t[[2]] + 1], clust2 = cds_var$paint[t[[1]] + 1],
               len = 0.5 * snp.dat$g - abs((pos1 - pos2) %% snp.dat$g - 0.5 * snp.dat$g), MI = t[[3]])
  }
  # lr_links.tsv (:362): options(ldwamd.native_tsv = TRUE) writes it from the device table with the library's threaded writer
  # (same bytes as write.table, 1e6 rows in ~0.3 s instead of seconds); the default is R's own write.table
  native_lr <- FALSE
  if (!perform_SR_analysis_only && !stream_lr && length(res[[2]][[3]]) > 0) {
    if (isTRUE(getOption("ldwamd.native_tsv", FALSE))) {
      .Call("ldwamd_write_links_tsv_begin", 1L, lr_save_path)   # r04: written by host threads while mergeNsort_sr_links runs below
      native_lr <- TRUE
      on.exit(if (native_lr) try(.Call("ldwamd_write_links_tsv_end"), silent = TRUE), add = TRUE)   # (an error below must not leave the writer unjoined)
    } else write.table(to_df(res[[2]]), file = lr_save_path, append = T, quote = F, row.names = F, col.names = F, sep = "\t")
  }
  rows_stay <- isTRUE(getOption("ldwamd.sr_rows_stay", FALSE)) && length(getOption("ldwamd.devices", 0L)) > 1 && !perform_SR_analysis_only
  if (rows_stay) {   # r05: the table is spread over the devices and never comes into R: the model and ARACNE run where the rows lie
    sr_links_red <- mergeNsort_sr_links_device(snp.dat, cds_var, sr_dist, srp_cutoff, runARACNE, plt_folder)
    if (native_lr) { .Call("ldwamd_write_links_tsv_end"); native_lr <- FALSE }
    if (!runARACNE) warning("ARACNE not run, all values will be set to 1")
    if (order_links) { sr_links_red <- sr_links_red[order(sr_links_red$srp_max, decreasing = T), ]; rownames(sr_links_red) <- NULL }
    write.table(x = sr_links_red, file = sr_save_path, append = T, quote = F, row.names = F, col.names = F, sep = "\t")
    return(sr_links_red)
  }
  sr <- to_df(res[[1]])
  sr_links <- lapply(1:cds_var$nclust, function(i) sr[sr$clust1 == i | sr$clust2 == i, ])
  sr_links_all <- mergeNsort_sr_links(cds_var = cds_var, sr_links = sr_links, sr_dist = sr_dist, plt_path = plt_folder,
                                      srp_cutoff = srp_cutoff)                # reference's own function, unchanged
  sr_links_red <- sr_links_all$sr_links_red
  chk <- sr_links_all$sr_links_ARACNE_check
  if (native_lr) { .Call("ldwamd_write_links_tsv_end"); native_lr <- FALSE }                # lr_links.tsv is complete from here on
  if (runARACNE) {
    sr_links_red$ARACNE <- as.numeric(.Call("ldwamd_aracne", sr_links_red$pos1, sr_links_red$pos2, sr_links_red$MI,
                                            chk$pos1, chk$pos2, chk$MI))
  } else {
    warning("ARACNE not run, all values will be set to 1"); sr_links_red$ARACNE <- 1
  }
  if (order_links) { sr_links_red <- sr_links_red[order(sr_links_red$srp_max, decreasing = T), ]; rownames(sr_links_red) <- NULL }
  write.table(x = sr_links_red, file = sr_save_path, append = T, quote = F, row.names = F, col.names = F, sep = "\t")
  sr_links_red
}

# mergeNsort_sr_links + runARACNE with the short-range table left on the GPU by ldwamd_mi_all_pairs
# (R/computePairwiseMI.R:400-495): the O(#links) work runs in HBM, R keeps fastLm and fitdist.  Same rows, order and
# columns as the reference's sr_links_red; use it instead of the to_df(res[[1]]) / mergeNsort_sr_links lines of
# perform_MI_computation above when the table is too large to bring into R (config 5: 2.25e9 rows).
mergeNsort_sr_links_device <- function(snp.dat, cds_var, sr_dist, srp_cutoff, runARACNE = TRUE, plt_path = NULL) {
  nclust <- cds_var$nclust
  S <- ceiling(sr_dist) - 1
  q <- .Call("ldwamd_sr_len_quantiles", as.integer(nclust), sr_dist, 0.95)
  byrow <- function(v) matrix(v, ncol = S, byrow = TRUE)
  qlo <- byrow(q[[1]]); qhi <- byrow(q[[2]]); cnt <- byrow(q[[3]])
  md <- matrix(NA_real_, nrow = nclust, ncol = S)
  for (ci in 1:nclust) {
    has <- which(cnt[ci, ] > 0)
    index <- 1 + (cnt[ci, has] - 1) * 0.95; h <- index - floor(index)           # stats::quantile type 7
    mx <- ifelse(h > 0 & qhi[ci, has] != qlo[ci, has], (1 - h) * qlo[ci, has] + h * qhi[ci, has], qlo[ci, has])
    mod <- RcppArmadillo::fastLm(cbind(log(has), 1), log(mx))                    # :428
    md[ci, seq_along(has)] <- exp(fitted(mod))                                   # looked up by the VALUE of len (:448)
    if (!is.null(plt_path))                                                      # the maxvls table of the reference (:422-439)
      saveRDS(object = data.frame(len = has, max = mx, fit = md[ci, seq_along(has)]), file = file.path(plt_path, paste("c", ci, "_fit_data.rds", sep = "")))
  }
  st <- matrix(.Call("ldwamd_sr_excess_stats", as.integer(nclust), as.numeric(t(md))), ncol = 5, byrow = TRUE)
  shape <- t(sapply(1:nclust, function(ci) {
    n <- st[ci, 1]; slx <- st[ci, 4]; sl1x <- st[ci, 5]; m <- st[ci, 2] / n; v <- st[ci, 3] / n - m^2
    aux <- m * (1 - m) / v - 1                                                   # fitdistrplus' moment start
    nll <- function(p) if (any(p <= 0)) Inf else -((p[1] - 1) * slx + (p[2] - 1) * sl1x - n * lbeta(p[1], p[2]))
    p <- optim(c(m * aux, (1 - m) * aux), nll, control = list(reltol = 1e-14))$par   # the beta likelihood only needs these sums
    c(p, lbeta(p[1], p[2]))
  }))
  r <- .Call("ldwamd_sr_pvalues_aracne", as.integer(nclust), as.numeric(t(md)), as.numeric(t(shape)), srp_cutoff, runARACNE)
  pos2 <- snp.dat$POS[r[[2]] + 1]; pos1 <- snp.dat$POS[r[[3]] + 1]   # (integer like the reference's MI_df columns)
  df <- data.frame(clust_c = r[[5]], pos1 = pos1, pos2 = pos2, clust1 = cds_var$paint[r[[3]] + 1], clust2 = cds_var$paint[r[[2]] + 1],
                   len = 0.5 * snp.dat$g - abs((pos1 - pos2) %% snp.dat$g - 0.5 * snp.dat$g), MI = r[[4]], srp_max = r[[8]],
                   ARACNE = r[[9]])
  df[order(r[[7]], ifelse(r[[7]], r[[6]], r[[5]]), r[[1]]), ]                    # reference row order (:470-486)
}

# Numeric core of analyse_long_range_links (R/lr_analyser.R:72-118) on the tables the device holds after
# perform_MI_computation(): lr_links_red with its ARACNE column, in the reference's order.
# sr_links: the reduced short-range links, i.e. what sr_links.tsv holds / perform_MI_computation() returned (R/lr_analyser.R:67)
analyse_long_range_links_device <- function(snp.dat, cds_var, sr_links, are_lrlinks_ordered = F) {
  ia <- match(sr_links$pos2, snp.dat$POS); ib <- match(sr_links$pos1, snp.dat$POS)
  if (anyNA(ia) || anyNA(ib)) stop("sr_links holds positions that are not in snp.dat$POS")   # NA_integer_ must not reach the device as an index
  r <- .Call("ldwamd_lr_tukey_aracne", 5000, as.integer(ia - 1L), as.integer(ib - 1L), as.numeric(sr_links$MI))
  if (r[[8]]) warning("Not enough lr links pass the Tukey criteria, ~5000 top links were retained instead")
  pos2 <- snp.dat$POS[r[[2]] + 1]; pos1 <- snp.dat$POS[r[[3]] + 1]   # (integer like the reference's MI_df columns)
  df <- data.frame(pos1 = pos1, pos2 = pos2, clust1 = cds_var$paint[r[[3]] + 1], clust2 = cds_var$paint[r[[2]] + 1],
                   len = 0.5 * snp.dat$g - abs((pos1 - pos2) %% snp.dat$g - 0.5 * snp.dat$g), MI = r[[4]], ARACNE = r[[5]])
  if (!are_lrlinks_ordered) { df <- df[order(df$MI, decreasing = T), ]; rownames(df) <- NULL }
  list(lr_links_red = df, thresholds = r[[7]])
}

# Numeric core of genomewide_LDMap (R/LDSummaryPlot.R:55-109): the matrix heatmap3 is given.
genomewide_LDMap_device <- function(reducer = NULL, from = NULL, to = NULL) {
  .Call("ldwamd_ldmap", if (is.null(reducer)) 0L else as.integer(round(reducer)), if (is.null(from)) 0L else as.integer(round(from)),
        if (is.null(to)) 0L else as.integer(round(to)))
}
