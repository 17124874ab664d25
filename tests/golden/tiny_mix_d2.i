e	a	1	1	1
e	h	2	1	1
e	b	1	1	1
e	c	2	1	1
e	d	2	1	1
h	f	1	1	2
h	g	2	1	2
