/* ORACLE (test infrastructure only) -- Jacobian short-Weierstrass (a=0) group law, instantiated twice
 * (G1 over Fq, G2 over Fq2) by field.c. Formulas: EFD dbl-2009-l, madd-2007-bl, add-2007-bl. */
void PFX(j_set_inf)(JT* r) { memset(r, 0, sizeof *r); FONE(&r->X); FONE(&r->Y); }
void PFX(j_from_affine)(JT* r, const AT* p) {
  if (p->inf) { PFX(j_set_inf)(r); return; }
  r->X = p->x; r->Y = p->y; FONE(&r->Z);
}
void PFX(j_dbl)(JT* r, const JT* p) {
  if (FISZERO(&p->Z)) { *r = *p; return; }
  FT A, B, C, D, E, F, t, X3, Y3, Z3;
  FSQR(&A, &p->X); FSQR(&B, &p->Y); FSQR(&C, &B);
  FADD(&t, &p->X, &B); FSQR(&t, &t); FSUB(&t, &t, &A); FSUB(&t, &t, &C); FADD(&D, &t, &t);
  FADD(&E, &A, &A); FADD(&E, &E, &A);
  FSQR(&F, &E);
  FSUB(&X3, &F, &D); FSUB(&X3, &X3, &D);
  FSUB(&t, &D, &X3); FMUL(&Y3, &E, &t);
  FADD(&t, &C, &C); FADD(&t, &t, &t); FADD(&t, &t, &t); FSUB(&Y3, &Y3, &t);
  FMUL(&Z3, &p->Y, &p->Z); FADD(&Z3, &Z3, &Z3);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
void PFX(j_add_affine)(JT* r, const JT* p, const AT* q) {
  if (q->inf) { *r = *p; return; }
  if (FISZERO(&p->Z)) { PFX(j_from_affine)(r, q); return; }
  FT Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
  FSQR(&Z1Z1, &p->Z);
  FMUL(&U2, &q->x, &Z1Z1);
  FMUL(&S2, &q->y, &p->Z); FMUL(&S2, &S2, &Z1Z1);
  FSUB(&H, &U2, &p->X);
  FSUB(&rr, &S2, &p->Y);
  if (FISZERO(&H)) {
    if (FISZERO(&rr)) { PFX(j_dbl)(r, p); } else { PFX(j_set_inf)(r); }
    return;
  }
  FADD(&rr, &rr, &rr);
  FSQR(&HH, &H);
  FADD(&I, &HH, &HH); FADD(&I, &I, &I);
  FMUL(&J, &H, &I);
  FMUL(&V, &p->X, &I);
  FSQR(&X3, &rr); FSUB(&X3, &X3, &J); FSUB(&X3, &X3, &V); FSUB(&X3, &X3, &V);
  FSUB(&t, &V, &X3); FMUL(&Y3, &rr, &t);
  FMUL(&t, &p->Y, &J); FADD(&t, &t, &t); FSUB(&Y3, &Y3, &t);
  FADD(&Z3, &p->Z, &H); FSQR(&Z3, &Z3); FSUB(&Z3, &Z3, &Z1Z1); FSUB(&Z3, &Z3, &HH);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
void PFX(j_add)(JT* r, const JT* p, const JT* q) {
  if (FISZERO(&q->Z)) { *r = *p; return; }
  if (FISZERO(&p->Z)) { *r = *q; return; }
  FT Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
  FSQR(&Z1Z1, &p->Z); FSQR(&Z2Z2, &q->Z);
  FMUL(&U1, &p->X, &Z2Z2); FMUL(&U2, &q->X, &Z1Z1);
  FMUL(&S1, &p->Y, &q->Z); FMUL(&S1, &S1, &Z2Z2);
  FMUL(&S2, &q->Y, &p->Z); FMUL(&S2, &S2, &Z1Z1);
  FSUB(&H, &U2, &U1);
  FSUB(&rr, &S2, &S1);
  if (FISZERO(&H)) {
    if (FISZERO(&rr)) { PFX(j_dbl)(r, p); } else { PFX(j_set_inf)(r); }
    return;
  }
  FADD(&rr, &rr, &rr);
  FADD(&I, &H, &H); FSQR(&I, &I);
  FMUL(&J, &H, &I);
  FMUL(&V, &U1, &I);
  FSQR(&X3, &rr); FSUB(&X3, &X3, &J); FSUB(&X3, &X3, &V); FSUB(&X3, &X3, &V);
  FSUB(&t, &V, &X3); FMUL(&Y3, &rr, &t);
  FMUL(&t, &S1, &J); FADD(&t, &t, &t); FSUB(&Y3, &Y3, &t);
  FADD(&Z3, &p->Z, &q->Z); FSQR(&Z3, &Z3); FSUB(&Z3, &Z3, &Z1Z1); FSUB(&Z3, &Z3, &Z2Z2); FMUL(&Z3, &Z3, &H);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
void PFX(j_to_affine)(AT* r, const JT* p) {
  if (FISZERO(&p->Z)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
  FT zi, zi2, zi3;
  FINV(&zi, &p->Z); FSQR(&zi2, &zi); FMUL(&zi3, &zi2, &zi);
  FMUL(&r->x, &p->X, &zi2); FMUL(&r->y, &p->Y, &zi3); r->inf = 0;
}
void PFX(_mul)(JT* r, const AT* p, const uint64_t k[4]) {
  JT acc; PFX(j_set_inf)(&acc);
  for (int i = 255; i >= 0; i--) {
    PFX(j_dbl)(&acc, &acc);
    if ((k[i / 64] >> (i % 64)) & 1) PFX(j_add_affine)(&acc, &acc, p);
  }
  *r = acc;
}
/* Montgomery's trick over the Z coordinates */
void PFX(_batch_to_affine)(AT* out, const JT* in, size_t n) {
  if (n == 0) return;
  FT* pre = (FT*)malloc(n * sizeof(FT));
  FT acc; FONE(&acc);
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!FISZERO(&in[i].Z)) FMUL(&acc, &acc, &in[i].Z);
  }
  FT ia; FINV(&ia, &acc);
  for (size_t i = n; i-- > 0;) {
    if (FISZERO(&in[i].Z)) { memset(&out[i], 0, sizeof(AT)); out[i].inf = 1; continue; }
    FT zi, zi2, zi3;
    FMUL(&zi, &ia, &pre[i]);
    FMUL(&ia, &ia, &in[i].Z);
    FSQR(&zi2, &zi); FMUL(&zi3, &zi2, &zi);
    FT x, y;
    FMUL(&x, &in[i].X, &zi2); FMUL(&y, &in[i].Y, &zi3);
    out[i].x = x; out[i].y = y; out[i].inf = 0;
  }
  free(pre);
}
